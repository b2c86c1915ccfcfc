"""Generate tests/golden/train_small.npz and tests/golden/ref_checkpoint_small.pth by running the REFERENCE's training step
(build container only).

    python oracle/gen_golden_train.py     # needs /root/reference; ~1 min of CPU

What runs is the reference's own code, unmodified, imported from a scratch copy of /root/reference:
  losses.get_step_fn (losses.py:151-210, train and eval branches), losses.optimization_manager (:38-52),
  losses.get_optimizer (:26-35), the three loss closures (:55-148), models/ema.ExponentialMovingAverage (ema.py:10-97),
  models/utils.create_model (utils.py:88-94: the model is wrapped in torch.nn.DataParallel, hence the `module.` keys), and
  the dict of utils.save_checkpoint (utils.py:22-29; the file itself imports tensorflow, which is absent here, so its four
  lines are the one thing written out below).
on the down-sized networks of tests/_util.TRAIN_CASES with seeded weights, dropout 0 and injected draws (torch.rand /
torch.randint / torch.randn_like patched while a step runs -- SURVEY F9), for TRAIN_STEPS optimisation steps (warm-up 2, so
lr = 0, lr/2, lr) followed by one eval step (EMA weights swapped in and out).

Stored per case: the loss of every step, the eval loss, for EVERY parameter tensor and step its L2 norm and the L2 norm of
its change since initialisation (parameters and EMA shadows, float64), and after the last step a probe set of tensors in
full (tests/_util.train_probe_names: one tensor of every leaf class).  The checkpoint case stores the reference-format checkpoint after two steps and the third
step's results, so a test can restore it into this repository's state and continue.

The script also asserts that oracle/train_oracle.py reproduces every stored number.  TEST INFRASTRUCTURE ONLY.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G                                  # noqa: E402  (sets sys.path for tests/ and the repo root)

import _util                                            # noqa: E402


def main():
    G.import_reference()
    from oracle import train_oracle
    import models.utils as ref_mutils            # noqa  (reference)
    import models.ncsnpp                         # noqa
    import models.ema as ref_ema                 # noqa
    import sde_lib as ref_sde_lib                # noqa
    import losses as ref_losses                  # noqa
    import ml_collections

    def ref_cfg_like(cfg):
        def conv(v):
            if hasattr(v, "items"):
                d = ml_collections.ConfigDict()
                for k, x in v.items():
                    d[k] = conv(x)
                return d
            return v
        return conv(cfg)

    torch.set_num_threads(min(16, os.cpu_count()))
    out = {}

    def run_case(name, case, ckpt_after=None):
        kind, _, sde_kind, continuous, reduce_mean, lw = case
        cfg = _util.train_case_config(case)
        cfg.device = torch.device("cpu")
        rcfg = ref_cfg_like(cfg)
        torch.manual_seed(0)
        model = ref_mutils.create_model(rcfg)                    # DataParallel(NCSNpp): `module.` state-dict keys
        sd = _util.fix_top_level_groupnorm(_util.seeded_state_dict(model.module, seed=1), model.module)
        missing = model.module.load_state_dict(sd, strict=False)
        assert set(missing.missing_keys) <= {"sigmas"} and not missing.unexpected_keys, missing
        full_sd = dict(sd); full_sd["sigmas"] = model.module.sigmas.clone()
        sde = _util.train_case_sde(ref_sde_lib, case, rcfg)
        optimizer = ref_losses.get_optimizer(rcfg, model.parameters())
        ema = ref_ema.ExponentialMovingAverage(model.parameters(), decay=rcfg.model.ema_rate)
        state = dict(optimizer=optimizer, model=model, ema=ema, step=0)
        optimize_fn = ref_losses.optimization_manager(rcfg)
        train_step = ref_losses.get_step_fn(sde, train=True, optimize_fn=optimize_fn, reduce_mean=reduce_mean,
                                            continuous=continuous, likelihood_weighting=lw)
        eval_step = ref_losses.get_step_fn(sde, train=False, optimize_fn=optimize_fn, reduce_mean=reduce_mean,
                                           continuous=continuous, likelihood_weighting=lw)
        names = [n for n, p in model.module.named_parameters() if p.requires_grad]
        init = {n: p.detach().clone() for n, p in model.module.named_parameters() if p.requires_grad}
        probes = _util.train_probe_names([(n, tuple(init[n].shape)) for n in names])
        sde_kwargs = dict(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales) \
            if sde_kind == "vesde" else dict(beta_min=cfg.model.beta_min, beta_max=cfg.model.beta_max, N=cfg.model.num_scales)
        orc = train_oracle.TrainState(cfg, full_sd, sde_kind, sde_kwargs, continuous, reduce_mean, lw)
        assert orc.names == names, "oracle's trainable set differs from the reference's"

        inputs = _util.train_case_inputs(name, cfg.model.num_scales, size=cfg.data.image_size)
        losses_, norms, ema_norms = [], [], []
        worst = 0.0
        for step in range(_util.TRAIN_STEPS):
            batch, u, labels, z = inputs[step]
            if ckpt_after is not None and step == ckpt_after:
                saved_state = {                                  # utils.py:23-28, verbatim field list
                    'optimizer': state['optimizer'].state_dict(),
                    'model': state['model'].state_dict(),
                    'ema': state['ema'].state_dict(),
                    'step': state['step'],
                }
                assert all(k.startswith("module.") for k in saved_state['model'])
                torch.save(saved_state, os.path.join(G.ROOT, "tests", "golden", "ref_checkpoint_small.pth"))
            with _util.inject_rng(u, labels, z):
                loss = train_step(state, batch)
            losses_.append(float(loss))
            o_loss = orc.train_step(batch, u, labels, z)
            assert abs(float(o_loss) - float(loss)) <= 1e-6 * abs(float(loss)), (name, step, float(o_loss), float(loss))
            cur = dict(model.module.named_parameters())
            norms.append([[float(cur[n].detach().double().norm()), float((cur[n].detach() - init[n]).double().norm())] for n in names])
            ema_norms.append([[float(s.double().norm()), float((s - init[n]).double().norm())] for s, n in zip(ema.shadow_params, names)])
            for n, s, os_ in zip(names, ema.shadow_params, orc.shadow):
                worst = max(worst, _util.rel_err(orc.params[n].detach(), cur[n].detach()), _util.rel_err(os_, s))
            if step == _util.TRAIN_STEPS - 1:
                for n in probes:
                    out["%s/p/%s" % (name, n)] = cur[n].detach().numpy().copy()
                    out["%s/e/%s" % (name, n)] = ema.shadow_params[names.index(n)].numpy().copy()
        assert state['step'] == _util.TRAIN_STEPS and ema.num_updates == _util.TRAIN_STEPS
        batch, u, labels, z = inputs[_util.TRAIN_STEPS]
        before = [p.detach().clone() for p in model.parameters()]
        with _util.inject_rng(u, labels, z):
            eval_loss = eval_step(state, batch)
        assert all(torch.equal(a, b.detach()) for a, b in zip(before, model.parameters()))
        o_eval = orc.eval_step(batch, u, labels, z)
        assert abs(float(o_eval) - float(eval_loss)) <= 1e-6 * abs(float(eval_loss)), (name, float(o_eval), float(eval_loss))
        print("%-18s losses %s eval %.6g | oracle-vs-reference worst parameter / shadow rel err %.3g (%d tensors, %d probes)"
              % (name, " ".join("%.6g" % v for v in losses_), float(eval_loss), worst, len(names), len(probes)))
        assert worst < 1e-6, worst
        out[name + "/loss"] = np.asarray(losses_, dtype=np.float64)
        out[name + "/eval_loss"] = np.asarray(float(eval_loss), dtype=np.float64)
        out[name + "/norms"] = np.asarray(norms, dtype=np.float64)
        out[name + "/ema_norms"] = np.asarray(ema_norms, dtype=np.float64)
        out[name + "/num_updates"] = np.asarray(ema.num_updates)

    for name, case in _util.TRAIN_CASES.items():
        run_case(name, case)
    run_case("ckpt", _util.TRAIN_CKPT_CASE, ckpt_after=2)
    path = os.path.join(G.ROOT, "tests", "golden", "train_small.npz")
    np.savez_compressed(path, **out)
    print("written:", path, "%.2f MB" % (os.path.getsize(path) / 1e6), "and ref_checkpoint_small.pth %.2f MB"
          % (os.path.getsize(os.path.join(G.ROOT, "tests", "golden", "ref_checkpoint_small.pth")) / 1e6))


if __name__ == "__main__":
    main()
