#!/usr/bin/env python3
"""First-contact check on a GPU box: libtdx.so built by hipcc 7.2 loads next to
torch's bundled HIP runtime, shares its streams/pointers, runs an elementwise
kernel bit-exactly, and the two peak probes give roofline denominators."""
import ctypes as C, os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "tiny_diffusion_amd", "libtdx.so"))
hips = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}
print("hip runtimes mapped:", hips)
assert torch.cuda.is_available()
dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)
P = C.c_void_p
st = torch.cuda.current_stream().cuda_stream
B = 64
x0 = torch.rand(B, 1, 28, 28, device=dev) * 2 - 1
nz = torch.randn(B, 1, 28, 28, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
betas = torch.linspace(1e-4, 0.02, 1000); ac = torch.cumprod(1 - betas, 0)
sa, sb = torch.sqrt(ac).to(dev), torch.sqrt(1.0 - ac).to(dev)
out = torch.empty_like(x0)
lib.tdx_q_sample.restype = C.c_int
lib.tdx_q_sample.argtypes = [P] * 6 + [C.c_int, C.c_int, P]
rc = lib.tdx_q_sample(x0.data_ptr(), nz.data_ptr(), t.data_ptr(), sa.data_ptr(), sb.data_ptr(), out.data_ptr(), B, 784, st)
torch.cuda.synchronize()
ref = sa[t].view(-1, 1, 1, 1) * x0 + sb[t].view(-1, 1, 1, 1) * nz
print("q_sample rc", rc, "bit-exact:", torch.equal(out, ref), "maxdiff", (out - ref).abs().max().item())

# fp32 MFMA peak probe
lib.tdx_probe_mfma_f32.restype = C.c_int
lib.tdx_probe_mfma_f32.argtypes = [P, C.c_int, C.c_int, P]
blocks, iters = 256 * 8, 20000
buf = torch.empty(blocks * 256, device=dev)
for _ in range(2):
    lib.tdx_probe_mfma_f32(buf.data_ptr(), iters, blocks, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); lib.tdx_probe_mfma_f32(buf.data_ptr(), iters, blocks, st); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
flop = blocks * 4 * iters * 4 * (2 * 32 * 32 * 2)
print(f"mfma f32 probe: {ms:.3f} ms -> {flop / ms / 1e9:.1f} TFLOP/s")

lib.tdx_probe_stream_copy.restype = C.c_int
lib.tdx_probe_stream_copy.argtypes = [P, P, C.c_int64, P]
n = 1 << 28  # 1 GiB each way
src = torch.empty(n, device=dev).normal_(); dst = torch.empty(n, device=dev)
for _ in range(2):
    lib.tdx_probe_stream_copy(src.data_ptr(), dst.data_ptr(), n, st)
torch.cuda.synchronize()
e0.record()
for _ in range(5):
    lib.tdx_probe_stream_copy(src.data_ptr(), dst.data_ptr(), n, st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"stream copy: {ms:.3f} ms -> {2 * n * 4 / ms / 1e9:.2f} TB/s (read+write)")
print("copy correct:", torch.equal(src, dst))
print(json.dumps({"cpu_count": os.cpu_count()}))
