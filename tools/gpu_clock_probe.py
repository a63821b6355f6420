#!/usr/bin/env python3
"""In-kernel shader clock of the fp32-MFMA convolution kernels vs the bare MFMA probe
(MI355X_MICROARCH.md, 'DVFS give-back' item 6): clock = d(s_memtime) / d(s_memrealtime) x 100 MHz,
stamped around the main loop of every workgroup (diagnostic knob conv_stamp; no product launch stamps),
after ~2 s of back-to-back launches on random data.  Answers: how much of the gap between the measured
TFLOP/s and the 157.3 TFLOP/s datasheet peak (2.4 GHz) is the clock the chip holds under this load?"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_diffusion_amd._lib import lib, check

dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
stamps = torch.zeros(8 * 65536, dtype=torch.int64, device=dev)
lib.tdx_diag_set_buffer(stamps.data_ptr(), stamps.numel() * stamps.element_size())
B = 256
out = {}


def clock_of(nwg):
    s = stamps.cpu()[: 8 * nwg].view(-1, 8).double()
    s = s[s[:, 1] > 0]
    mhz = (s[:, 0] / s[:, 1] * 100.0)
    tl = None
    if s[:, 4].max() > 0:   # timeline of the last stamped launch, in us (10-ns ticks)
        t0 = s[:, 2].min()
        tl = {"first_loop_start_us": 0.0, "last_loop_start_us": float((s[:, 2].max() - t0) / 100),
              "loop_us_min_med_max": [float(v / 100) for v in (s[:, 1].min(), s[:, 1].median(), s[:, 1].max())],
              "first_loop_end_us": float((s[:, 3].min() - t0) / 100), "last_loop_end_us": float((s[:, 3].max() - t0) / 100),
              "epilogue_us_min_med_max": [float(v / 100) for v in ((s[:, 4] - s[:, 3]).min(), (s[:, 4] - s[:, 3]).median(),
                                                                   (s[:, 4] - s[:, 3]).max())],
              "last_epilogue_end_us": float((s[:, 4].max() - t0) / 100), "workgroups": int(s.shape[0])}
    return float(mhz.median()), float(mhz.min()), float(mhz.max()), float(s[:, 0].median()), tl


def run(name, fn, nwg, flop, secs=2.0):
    lib.tdx_tune_set(b"conv_stamp", 0)
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:        # warm the chip: the clock settles under sustained load
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); n += 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    stamps.zero_()
    lib.tdx_tune_set(b"conv_stamp", 1)
    fn()
    torch.cuda.synchronize()
    lib.tdx_tune_set(b"conv_stamp", 0)
    med, lo, hi, cyc, tl = clock_of(nwg)
    tf = flop / ms / 1e9
    peak_at_clock = 157.3 * med / 2400.0
    out[name] = {"ms": round(ms, 4), "tflops": round(tf, 1), "clock_mhz_median": round(med), "clock_mhz_min": round(lo),
                 "clock_mhz_max": round(hi), "peak_at_that_clock_tflops": round(peak_at_clock, 1),
                 "frac_of_peak_at_that_clock": round(tf / peak_at_clock, 3), "main_loop_cycles_median": round(cyc),
                 "timeline": tl}
    print(name, out[name], flush=True)


# bare MFMA probe (one wave per SIMD, four accumulator chains)
blocks, iters = 1024, 8192
pout = torch.empty(blocks * 256, device=dev)
run("mfma_probe", lambda: check(lib.tdx_probe_mfma_f32(pout.data_ptr(), iters, blocks, st)), blocks,
    blocks * 4 * iters * 4 * 4096.0)
DBG = int(os.environ.get("TDX_CONV_DBG", "0"))
lib.tdx_tune_set(b"conv_dbg", DBG)
for cin, cout, H in ((512, 128, 16), (1024, 256, 8), (128, 128, 28), (256, 256, 14), (512, 512, 7), (64, 64, 32)):
    M = B * H * H
    x = torch.randn(M * cin, device=dev)
    wf = torch.randn(cout * 9 * cin, device=dev) * 0.02
    y = torch.empty(M * cout, device=dev)
    bias = torch.zeros(cout, device=dev)
    tiles = lib.tdx_conv3x3_stat_tiles(B, H, H, cin, cout)
    stat = torch.empty(tiles * 2 * cout, device=dev)
    shape = lib.tdx_conv3x3_tile_shape(B, H, H, cin, cout, 0)
    bm, bn = shape // 1000, shape % 1000
    nwg = ((M + bm - 1) // bm + 7) // 8 * 8 * (cout // bn)
    run(f"conv_fwd_{cin}_{cout}_{H} tile {bm}x{bn}",
        lambda: check(lib.tdx_conv3x3_fwd(x.data_ptr(), wf.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, H, cin, cout, 4,
                                          None, None, None, None, stat.data_ptr(), st)),
        nwg, 2.0 * M * 9 * cin * cout)
lib.tdx_diag_set_buffer(None, 0)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/clock_probe.json", "w"), indent=1)
