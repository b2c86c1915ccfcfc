"""torch.autograd bridge: NCSNpp.forward under grad mode.

User code written against the reference (custom losses, likelihood.py's Hutchinson divergence,
controllable generation) differentiates through `model(x, labels)` with torch autograd.  This Function
runs the HIP forward program (train-mode engine: activations stay resident) and, on `.backward()`, the
HIP backward program (backward.TrainEngine); parameter gradients come back as the autograd outputs for
the parameters, d/dx when the input requires grad.  The stock DSM training step does not go through
here -- losses.get_step_fn runs the fused loss head / backward / optimizer directly.
"""
import torch

from . import backward as B


class _UNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, seed, x, cond, *params):
        ctx.eng = eng
        ctx.need_x = x.requires_grad
        y = eng.forward_train(x.detach(), cond.detach(), seed=seed)
        ctx.gen = eng._fwd_generation = getattr(eng, "_fwd_generation", 0) + 1
        return y.clone()

    @staticmethod
    def backward(ctx, gy):
        eng = ctx.eng
        if getattr(eng, "_fwd_generation", 0) != ctx.gen:
            raise RuntimeError("score_sde_pytorch_amd: backward through a NCSNpp forward whose activations were "
                               "overwritten by a later forward of the same shape (one live graph per engine)")
        eng.backward(gy.contiguous())
        n_params = len(ctx.needs_input_grad) - 4
        grads = tuple(eng.flat.grad_view(p).clone() for p in eng.flat.params) if eng.param_grads else (None,) * n_params
        gx = eng.gx_view().clone() if ctx.need_x else None
        return (None, None, gx, None) + grads


def unet_apply(model, x, cond):
    param_grads = any(p.requires_grad for p in B.trainable_params(model))
    key = ("train", x.device.index, x.shape[0], x.shape[2], x.shape[3], bool(model.training), bool(x.requires_grad), param_grads)
    eng = model._engines.get(key)
    if eng is None:
        eng = B.TrainEngine(model, x.shape[0], x.shape[2], x.shape[3], x.device, dropout=model.training,
                            input_grad=x.requires_grad, param_grads=param_grads)
        model._engines[key] = eng
    model._dropout_calls = getattr(model, "_dropout_calls", 0) + 1
    seed = (int(torch.initial_seed()) * 1000003 + model._dropout_calls) & 0x7FFFFFFF
    return _UNetFunction.apply(eng, seed, x, cond, *(eng.flat.params if param_grads else ()))
