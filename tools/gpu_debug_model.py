"""Per-layer parity dump of the HIP plan against the oracle on the GPU box.
usage: python tools/gpu_debug_model.py <cfg-name> [fp32|bf16] [train|eval] [bwd]"""
import os
import sys
os.environ["DYK_DEBUG_PLAN"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "double-yolo-kaist_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from build_utils.parse_config import materialize_cfg  # noqa: E402
from helpers import oracle_net  # noqa: E402
from models import YOLO  # noqa: E402


def tref_to_nchw(plan, t):
    es = t.esize
    dt = torch.float32 if es == 4 else torch.bfloat16
    a = plan.arenas[t.arena].tensor
    n = t.npix * t.ld
    flat = a[t.off:t.off + n * es].view(dt)[:n] if False else a[t.off:t.off + (n - (t.ld - t.C)) * es].view(dt)
    # build strided view [B,H,W,C]
    v = torch.as_strided(flat, (t.B, t.H, t.W, t.C), (t.H * t.W * t.ld, t.W * t.ld, t.ld, 1))
    return v.float().permute(0, 3, 1, 2).contiguous().cpu()


def main():
    name = sys.argv[1]
    dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    mode = sys.argv[3] if len(sys.argv) > 3 else "eval"
    bwd = "bwd" in sys.argv
    torch.manual_seed(0)
    net = oracle_net(name)
    sd = net.synth_state(0)
    model = YOLO(materialize_cfg(name))
    model.load_state_dict(sd)
    model.dyk_dtype = dtype
    model = model.cuda()
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(2, 3, 128, 160, generator=g)
    y = torch.rand(2, 3, 128, 160, generator=g)
    training = mode == "train"
    model.train(training)
    if training:
        for k, v in sd.items():
            if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
    (ref, every) = net.forward(sd, x, y, training=training, keep_all=True)
    if training and bwd:
        for t in every:
            if t.requires_grad and not t.is_leaf:
                t.retain_grad()
    out = model(x.cuda(), y.cuda())
    torch.cuda.synchronize()
    plan = list(model.engine.plans.values())[0]
    worst = 0.0
    for i, (t, r) in enumerate(zip(plan.outs, every)):
        if t is None or plan.info[i]["kind"] == "yolo":
            continue
        got = tref_to_nchw(plan, t)
        r = r.detach()
        if got.shape != r.shape:
            print("layer %3d %-12s SHAPE %s vs %s" % (i, plan.info[i]["kind"], tuple(got.shape), tuple(r.shape)))
            continue
        err = (got - r).abs().max().item()
        rel = err / max(r.abs().max().item(), 1e-6)
        flag = " <<<<" if rel > (1e-3 if dtype == "fp32" else 8e-2) else ""
        if flag or i % 20 == 0 or 236 <= i <= 262:
            print("layer %3d %-12s max|ref| %.3e err %.3e rel %.2e%s" % (i, plan.info[i]["kind"], r.abs().max().item(), err, rel, flag))
        worst = max(worst, rel)
    print("worst layer rel err %.3e" % worst)
    if training:
        for i, (o, r) in enumerate(zip(out, ref)):
            print("head %d err %.3e (max %.3e)" % (i, (o.detach().cpu() - r.detach()).abs().max().item(), r.abs().max().item()))
    else:
        io, p = out
        print("io err %.3e (max %.3e)" % ((io.cpu() - ref[0]).abs().max().item(), ref[0].abs().max().item()))
    if training and bwd:
        loss_ref = sum((t ** 2).mean() for t in ref)
        loss_ref.backward()
        loss = sum((t ** 2).mean() for t in out)
        loss.backward()
        torch.cuda.synchronize()
        print("loss ref %.6f got %.6f" % (loss_ref.item(), loss.item()))
        import ctypes
        for i in (279, 278, 268, 257, 250):
            rec = plan.info[i]
            if rec["kind"] != "conv" or not rec["bn"]:
                continue
            yr = tref_to_nchw(plan, rec["y_raw"])
            ref_raw = net.raw[i].detach()
            cout = rec["cout"]
            ws = plan.arenas["ws"].tensor
            v = ws[rec["vecs"]:rec["vecs"] + 16 * cout].view(torch.float32).cpu().view(4, cout)
            mean_ref = ref_raw.mean((0, 2, 3))
            var_ref = ref_raw.var((0, 2, 3), unbiased=False)
            print("layer %d y_raw err %.3e (max %.3e) mean err %.3e rstd err %.3e (max rstd %.3e)" % (
                i, (yr - ref_raw).abs().max().item(), ref_raw.abs().max().item(), (v[2] - mean_ref).abs().max().item(),
                (v[3] - 1 / torch.sqrt(var_ref + 1e-5)).abs().max().item(), v[3].abs().max().item()))
        for i in range(len(plan.outs) - 1, -1, -1):
            t = plan.outs[i]
            kind = plan.info[i]["kind"]
            if t is None or kind in ("yolo",) or plan.info[i].get("alias") or t.tid not in plan.grads:
                continue
            gr = every[i].grad
            if gr is None:
                continue
            got = tref_to_nchw(plan, plan.grads[t.tid])
            err = (got - gr).abs().max().item()
            rel = err / max(gr.abs().max().item(), 1e-12)
            print("dL/d(out %3d %-9s) max|ref| %.3e err %.3e rel %.2e%s" % (i, kind, gr.abs().max().item(), err, rel, " <<<<" if rel > 1e-2 else ""))
        bad = 0
        for k, p in model.named_parameters():
            gr = sd[k].grad
            gg = p.grad.detach().cpu()
            err = (gg - gr).abs().max().item()
            rel = err / max(gr.abs().max().item(), 1e-8)
            tol = 2e-3 if dtype == "fp32" else 1.5e-1
            li = int(k.split(".")[1])
            if rel > tol:
                bad += 1
            if li >= 268 or li < 3:
                print("GRAD %-45s max|ref| %.3e err %.3e rel %.2e %s" % (k, gr.abs().max().item(), err, rel, "BAD" if rel > tol else "ok"))
        print("params with grad mismatch: %d of %d" % (bad, len(list(model.parameters()))))
        # running stats
        sdm = model.state_dict()
        rs = max((sdm[k].cpu() - v).abs().max().item() for k, v in sd.items() if k.endswith(("running_mean", "running_var")))
        print("running stats max err %.3e" % rs)


if __name__ == "__main__":
    main()
