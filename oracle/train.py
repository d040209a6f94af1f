"""CPU restatement of the reference's training step (TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product path).

  loss_and_grads   GaussianDiffusion.forward / p_losses (src/hicdiff.py:711-755, src/hicdiff_condition.py:715-750) followed by
                   loss.backward() (train.py:131-132): the oracle's functional eps-net under torch autograd
  adam_step        torch.optim.Adam(params, lr=2e-5) as train.py:111 builds it: betas (0.9, 0.999), eps 1e-8, no weight decay,
                   no amsgrad; m <- b1 m + (1-b1) g; v <- b2 v + (1-b2) g^2; p <- p - lr/(1-b1^k) * m / (sqrt(v)/sqrt(1-b2^k) + eps)
"""
import math

import torch

from . import nets as ON


def loss_and_grads(sd, cfg, buf, x0, t, eps, cond=None, loss_type="l2", objective="pred_noise"):
    """t: int64 timesteps; for SR3 nets (cfg.sr3) the continuous noise level as a float tensor (src/hicdiff_sr3.py:750-792:
    x_t = level x0 + sqrt(1 - level^2) eps, plain mean reduction)."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    fn = ON.unet_eps if isinstance(cfg, ON.UnetCfg) else ON.hicedrn_eps
    if cfg.sr3:
        lv = t.float().reshape(-1, 1, 1, 1)
        out = fn(p, lv * x0 + (1 - lv ** 2).sqrt() * eps, t.float().reshape(-1, 1), cond, cfg)
        loss = ((out - eps).abs() if loss_type == "l1" else (out - eps) ** 2).mean()
    else:
        a = buf["sqrt_alphas_cumprod"].gather(-1, t).reshape(-1, 1, 1, 1)
        s = buf["sqrt_one_minus_alphas_cumprod"].gather(-1, t).reshape(-1, 1, 1, 1)
        out = fn(p, a * x0 + s * eps, t, cond, cfg)
        target = eps if objective == "pred_noise" else x0 if objective == "pred_x0" else a * eps - s * x0      # src/hicdiff.py:733-741, predict_v :542-546
        per = (out - target).abs() if loss_type == "l1" else (out - target) ** 2
        loss = (per.reshape(per.shape[0], -1).mean(dim=1) * buf["p2_loss_weight"].gather(-1, t)).mean()
    loss.backward()
    return loss.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}


def adam_step(params, grads, m, v, step, lr=2e-5, b1=0.9, b2=0.999, eps=1e-8):
    """In place on the dicts' tensors; `step` counts from 1."""
    c1, c2 = 1 - b1 ** step, 1 - b2 ** step
    for k, p in params.items():
        g = grads[k]
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v[k].sqrt() / math.sqrt(c2)).add_(eps)
        p.addcdiv_(m[k], denom, value=-lr / c1)


def sample_of(tensor, n=96):
    """The fixed subset of a gradient / parameter tensor the fixtures keep: n entries at a stride coprime to typical shapes."""
    flat = tensor.detach().reshape(-1)
    idx = (torch.arange(n, dtype=torch.int64) * 7919) % flat.numel()
    return flat[idx].clone()
