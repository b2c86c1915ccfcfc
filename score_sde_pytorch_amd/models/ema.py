"""Exponential moving average of parameters (drop-in for the reference's models/ema.py:10-97).

Same constructor, `update / copy_to / store / restore / state_dict / load_state_dict` and the same
warm-up rule decay = min(decay, (1 + n) / (10 + n)).  The shadow parameters can be re-homed into one
flat buffer laid out like `backward.FlatParams` (`flatten_like`), after which the fused optimizer
kernel (libssde_hip: ssde_adam_clip_ema) updates them in the same pass as Adam; `update()` then only
advances the counter when told the device already did the arithmetic.
"""
import torch


def _touch(param):
    """A parameter that is a view of a flat buffer (backward.FlatParams) also advances that buffer's generation."""
    flat = getattr(param, "_ssde_flat", None)
    if flat is not None:
        flat.touch()


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError('Decay must be between 0 and 1')
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.clone().detach() for p in parameters if p.requires_grad]
        self.collected_params = []
        self._flat = None
        self._flat_owner = None
        self._stored_flat = None

    # -- models/ema.py:32-51
    def next_decay(self):
        """Advance the update counter and return this update's decay (models/ema.py:44-47)."""
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        return decay

    def update(self, parameters):
        one_minus_decay = 1.0 - self.next_decay()
        with torch.no_grad():
            parameters = [p for p in parameters if p.requires_grad]
            for s_param, param in zip(self.shadow_params, parameters):
                s_param.sub_(one_minus_decay * (s_param - param))

    def _flat_home(self, parameters):
        """The FlatParams whose buffer these parameters are views of, when the shadow copy lives in a twin buffer."""
        flat = self._flat_owner
        if flat is None or self._flat is None:
            return None, parameters
        parameters = list(parameters)
        mine = [p for p in parameters if p.requires_grad]
        if len(mine) != len(flat.model_order) or any(a is not b for a, b in zip(mine, flat.model_order)) or not flat.owns_params(mine):
            return None, parameters
        return flat, parameters

    def copy_to(self, parameters):
        flat, parameters = self._flat_home(parameters)
        if flat is not None:            # one device copy instead of one per tensor (models/ema.py:53-63)
            flat.data.copy_(self._flat)
            flat.touch()
            return
        parameters = [p for p in parameters if p.requires_grad]
        # `param.copy_` (not `param.data.copy_`) bumps Tensor._version: engine.WeightStore.refresh keys the re-pack of
        # its kernel-layout weight copies on it, so an engine / captured sampler graph lowered before the swap sees it
        with torch.no_grad():
            for s_param, param in zip(self.shadow_params, parameters):
                if param.requires_grad:
                    param.copy_(s_param.data)
                    _touch(param)

    def store(self, parameters):
        flat, parameters = self._flat_home(parameters)
        if flat is not None:            # models/ema.py:65-73; parameters outside the flat buffer are not trainable and
            self._stored_flat = flat.data.clone()    # copy_to never touches them
            self.collected_params = []
            return
        self._stored_flat = None
        self.collected_params = [param.clone() for param in parameters]

    def restore(self, parameters):
        flat, parameters = self._flat_home(parameters)
        if flat is not None and self._stored_flat is not None:
            flat.data.copy_(self._stored_flat)
            flat.touch()
            return
        with torch.no_grad():
            for c_param, param in zip(self.collected_params, parameters):
                param.copy_(c_param.data)
                _touch(param)

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)

    def load_state_dict(self, state_dict):
        self.decay = state_dict['decay']
        self.num_updates = state_dict['num_updates']
        with torch.no_grad():
            for s, v in zip(self.shadow_params, state_dict['shadow_params']):
                s.copy_(v.to(s.device))          # in place: keeps a flat re-homing valid

    # -- fused path
    def flatten_like(self, flat):
        """Re-home the shadow parameters into one buffer with the layout of `flat` (backward.FlatParams)."""
        if self._flat is not None and self._flat_owner is flat:
            return self._flat
        buf = torch.zeros(flat.numel, dtype=torch.float32, device=flat.data.device)
        assert len(self.shadow_params) == len(flat.model_order)
        new = []
        for s, p in zip(self.shadow_params, flat.model_order):      # shadow parameters follow model.parameters() order
            o, n = flat.index[id(p)]
            buf[o:o + n].copy_(s.reshape(-1).to(buf.device, torch.float32))
            new.append(buf[o:o + n].view(p.shape))
        self.shadow_params = new
        self._flat = buf
        self._flat_owner = flat
        return buf
