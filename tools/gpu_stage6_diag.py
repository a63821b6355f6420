#!/usr/bin/env python3
"""Round-2 diagnosis of the round-1 `time_stage=6` anomaly (wrong time_embedding.0.weight gradient in
some workgroups).  Runs the round-1 reproducer (fresh models, an old plan destroyed
while the new one runs) with the ORIGINAL kernel (knob time_l1_impl=1: int64 t read from the
workspace copy made by hipMemcpyAsync) and, for every mismatching iteration, recomputes dW1 on the
host from the tensors the kernel read (g_h, pre) with
    (a) the t of THIS iteration, (b) the t of earlier iterations (a stale read of the address),
and reports which one reproduces the wrong 32-column groups.  Second arm: the same binary with the
forward's input copies done by a copy KERNEL instead of hipMemcpyAsync (knob input_copy=1) - a
host-side change only, so "any change to the kernel hides it" does not apply.

    python tools/gpu_stage6_diag.py [B] [iters]
"""
import gc, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiny_diffusion_amd._lib as L
from tiny_diffusion_amd.diffusion import NoiseModel

lib = L.lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 80
KEY = "time_embedding.0.weight"


def fresh():
    torch.manual_seed(1234)
    return NoiseModel().cuda().train()


def grads(m, x, t):
    (m(x, t) ** 2).mean().backward()


def silu_grad(x):
    s = torch.sigmoid(x)
    return s * (1 + x * (1 - s))


def host_dw1(plan, tvals):
    g_h = plan.tensor("time_g_h").view(B, 256).double().cpu()
    pre = plan.tensor("pre").view(B, 256).double().cpu()
    gp = g_h * silu_grad(pre)
    return (gp * tvals.double().view(B, 1)).sum(0), gp.sum(0)


def arm(name, l1_impl, input_copy, stage, destroy_while_running=True):
    lib.tdx_tune_set(b"time_l1_impl", l1_impl)
    lib.tdx_tune_set(b"input_copy", input_copy)
    bad, report, t_hist = 0, [], []
    for it in range(iters):
        g = torch.Generator(device="cuda").manual_seed(it)
        x = torch.randn(B, 1, 28, 28, device="cuda", generator=g)
        t = torch.randint(0, 1000, (B,), device="cuda", generator=g)
        lib.tdx_tune_set(b"time_stage_diag", 14)
        ref_m = fresh(); grads(ref_m, x, t); torch.cuda.synchronize()
        ref = {k: p.grad.clone() for k, p in ref_m.named_parameters()}
        del ref_m; gc.collect()
        lib.tdx_tune_set(b"time_stage_diag", stage)
        a = fresh(); grads(a, x, t); torch.cuda.synchronize()
        b = fresh()
        grads(b, x, t)
        if destroy_while_running:
            del a; gc.collect()
            torch.cuda.synchronize()
        else:
            torch.cuda.synchronize()
            del a; gc.collect()
        wrong = [k for k, p in b.named_parameters() if not torch.equal(p.grad, ref[k])]
        if wrong:
            bad += 1
            plan = list(b._plans.values())[0]
            got = dict(b.named_parameters())[KEY].grad.view(-1).double().cpu()
            want = ref[KEY].view(-1).double().cpu()
            cols = (got != want).nonzero().view(-1).tolist()
            groups = sorted({c // 32 for c in cols})
            t_copy = plan.tensor("t_copy").view(torch.int64)[:B].cpu()
            entry = {"iter": it, "wrong_params": wrong, "wrong_groups": groups, "n_wrong_cols": len(cols),
                     "t_copy_equals_t": bool(torch.equal(t_copy, t.cpu())),
                     "tf_equals_t": bool(torch.equal(plan.tensor("tf")[:B].cpu(), t.cpu().float()))}
            if cols:
                cur, _ = host_dw1(plan, t.cpu())
                scale = want.abs().max().item()
                entry["host_with_current_t_matches_REFERENCE"] = float((cur - want).abs().max() / scale)
                entry["host_with_current_t_vs_GOT_on_wrong_cols"] = float((cur[cols] - got[cols]).abs().max() / scale)
                cands = {}
                for back, told in enumerate(reversed(t_hist[-6:]), 1):
                    old, _ = host_dw1(plan, told)
                    cands[f"t_of_iter-{back}"] = float((old[cols] - got[cols]).abs().max() / scale)
                entry["host_with_stale_t_vs_GOT_on_wrong_cols"] = cands
                # any per-sample t' that explains got?  solve least squares gp @ t' = got on the wrong cols
                g_h = plan.tensor("time_g_h").view(B, 256).double().cpu()
                pre = plan.tensor("pre").view(B, 256).double().cpu()
                gp = (g_h * silu_grad(pre))[:, cols]              # (B, ncols)
                if len(cols) >= B:
                    sol = torch.linalg.lstsq(gp.t(), got[cols].unsqueeze(1)).solution.view(-1)
                    resid = (gp.t() @ sol - got[cols]).abs().max().item() / scale
                    entry["lstsq_t_explaining_got"] = {"resid": resid, "t_fit_first8": sol[:8].tolist(),
                                                       "t_true_first8": t.cpu()[:8].tolist(),
                                                       "n_samples_differing": int(((sol - t.cpu().double()).abs() > 0.5).sum())}
                # a stale CACHE LINE of t would corrupt 8 (64 B) or 16 (128 B) consecutive, aligned samples: fit the
                # error with a per-sample offset restricted to every such window and keep the one that explains it
                import struct
                diff = (got - want)[cols]
                gp_all = g_h * silu_grad(pre)
                best = None
                for win in (8, 16):
                    for st0 in range(0, B, win):
                        A = gp_all[st0:st0 + win][:, cols].t()           # (ncols, win)
                        sol = torch.linalg.lstsq(A, diff.unsqueeze(1)).solution.view(-1)
                        resid = float((A @ sol - diff).abs().max() / max(diff.abs().max().item(), 1e-30))
                        if best is None or resid < best["resid"]:
                            best = {"resid": resid, "win": win, "start": st0, "delta": sol.tolist()}
                if best is not None:
                    tb = [int(round(float(t[best["start"] + i]) + d)) for i, d in enumerate(best["delta"])]
                    def as_f32(u):
                        return struct.unpack("<f", struct.pack("<I", u & 0xffffffff))[0]
                    best["t_used"] = tb
                    best["t_true"] = [int(v) for v in t[best["start"]:best["start"] + best["win"]].tolist()]
                    best["t_used_hex"] = [hex(v & 0xffffffffffffffff) for v in tb]
                    best["lo_dword_as_f32"] = [as_f32(v) for v in tb]
                    best["hi_dword_as_f32"] = [as_f32(v >> 32) for v in tb]
                    del best["delta"]
                    entry["stale_line_fit"] = best
            if l1_impl == 2:
                entry["kernel_saw"] = DUMP(t.cpu())
            report.append(entry)
            print(f"[{name}] iter {it}: BAD {json.dumps(entry)}", flush=True)
        elif l1_impl == 2 and it == 1:
            print(f"[{name}] iter {it}: ok; kernel saw {json.dumps(DUMP(t.cpu()))}", flush=True)
        elif it % 20 == 0:
            print(f"[{name}] iter {it}: ok", flush=True)
        t_hist.append(t.cpu())
        del b; gc.collect()
    print(f"[{name}] done: {bad} bad of {iters}", flush=True)
    return {"arm": name, "bad": bad, "iters": iters, "report": report}


# instrumented variant first: what did the kernel actually load?
dbg = torch.zeros(64 * 8 + 8 * 2048, dtype=torch.int32, device="cuda")
lib.tdx_diag_set_buffer(dbg.data_ptr(), dbg.numel() * dbg.element_size())


def dump_seen(t_true):
    d = dbg.cpu()
    hdr = d[:64].view(8, 8)
    seen = d[64 * 8:].view(8, 1024, 2)[:, :B]
    vals = (seen[..., 0].to(torch.int64) & 0xffffffff) | (seen[..., 1].to(torch.int64) << 32)
    out = {"t_ptr": [hex(((int(h[1]) & 0xffffffff) << 32) | (int(h[0]) & 0xffffffff)) for h in hdr],
           "xcc": [int(h[2]) for h in hdr], "t_start": [int(h[3]) & 0xffffffff for h in hdr],
           "t_end": [int(h[4]) & 0xffffffff for h in hdr], "B_seen": [int(h[5]) for h in hdr], "wrong": []}
    for wg in range(8):
        bad_n = (vals[wg] != t_true).nonzero().view(-1).tolist()
        if bad_n:
            out["wrong"].append({"wg": wg, "n": bad_n[:16], "count": len(bad_n),
                                 "seen_hex": [hex(int(vals[wg, n]) & 0xffffffffffffffff) for n in bad_n[:16]],
                                 "true": [int(t_true[n]) for n in bad_n[:16]]})
    return out


DUMP = dump_seen
out = ([] if os.environ.get("TDX_DIAG_SKIP_INSTRUMENTED") else [arm("instrumented old kernel, stage 6", 2, 0, 6)]) + [
    arm("old-kernel+hipMemcpyAsync, stage 6", 1, 0, 6)] + (
    [arm("old kernel with agent-scope atomic loads of t, stage 6", 3, 0, 6)] if os.environ.get("TDX_DIAG_COHERENT") else [])
lib.tdx_diag_set_buffer(None, 0); lib.tdx_tune_set(b"time_stage_diag", 14); lib.tdx_tune_set(b"time_l1_impl", 0); lib.tdx_tune_set(b"input_copy", 0)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/stage6_diag.json", "w"), indent=1)
print(json.dumps([{k: v for k, v in o.items() if k != "report"} for o in out]))
