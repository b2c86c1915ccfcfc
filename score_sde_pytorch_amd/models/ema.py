"""Exponential moving average of parameters with warm-up decay min(d, (1+n)/(10+n)).

Same interface as the reference's models/ema.py:10-97 (`update`, `copy_to`, `store`,
`restore`, `state_dict`, `load_state_dict`; `shadow_params` is a python list of tensors,
which is what reference checkpoints hold).  On GPU the whole update is one multi-tensor
launch (`torch._foreach_*`) instead of a python loop over 572 tensors.
"""
import torch


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError('Decay must be between 0 and 1')
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.clone().detach() for p in parameters if p.requires_grad]
        self.collected_params = []

    def update(self, parameters):
        d = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            d = min(d, (1 + self.num_updates) / (10 + self.num_updates))
        live = [p.detach() for p in parameters if p.requires_grad]
        with torch.no_grad():
            # s <- s - (1 - d) * (s - p)
            diff = torch._foreach_sub(self.shadow_params, live)
            torch._foreach_add_(self.shadow_params, diff, alpha=-(1.0 - d))

    def copy_to(self, parameters):
        live = [p for p in parameters if p.requires_grad]
        with torch.no_grad():
            for s, p in zip(self.shadow_params, live):
                p.data.copy_(s.data)

    def store(self, parameters):
        self.collected_params = [p.clone() for p in parameters]

    def restore(self, parameters):
        for c, p in zip(self.collected_params, parameters):
            p.data.copy_(c.data)

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)

    def load_state_dict(self, state_dict):
        self.decay = state_dict['decay']
        self.num_updates = state_dict['num_updates']
        self.shadow_params = state_dict['shadow_params']
