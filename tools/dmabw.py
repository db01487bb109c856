"""LDS-DMA ring microbenchmark (no product code): read bandwidth of the access pattern k_pool / k_dynconv use
(1 KiB wave-instructions = 8 channel rows x 128 B, rows 64 KB apart in NCHW) as a function of the tile layout, and
with dummy per-tile MFMA / dependent LDS-read work.  Builds tools/dmabw.hip with hipcc on the box it runs on."""
import ctypes as C, os, subprocess, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from bench import time_op
so = os.path.join(HERE, "libdmabw.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                       os.path.join(HERE, "dmabw.hip"), "-o", so])
lib = C.CDLL(so)
lib.dmabw.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
B, HWp = 24, 32768
buf = torch.randint(0, 2**31 - 1, (B * 256 * HWp // 2,), dtype=torch.int32, device="cuda")   # 403 MB = one bf16 feature map x 24 frames
out = torch.zeros(4, dtype=torch.int32, device="cuda")
def run(name, row_stride, tile_stride, tiles_per_wg, wg_stride, wgs, nbuf, nmfma=0, nlds=0):
    t = time_op(lambda: lib.dmabw(buf.data_ptr(), row_stride, tile_stride, tiles_per_wg, wg_stride, wgs, nbuf, out.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream, nmfma, nlds), 10)
    print(f"{name}: {t*1e3:.1f} us = {wgs*tiles_per_wg*32768/t/1e9:.2f} TB/s")
for nbuf, per_cu in ((4, 1), (2, 2)):
    wgs = 256 * per_cu
    tpw = B * 512 // wgs
    run(f"ring {nbuf} x 32 KiB, {per_cu} WG/CU | rows 1.5 MB apart, contiguous tile range per WG", B * 65536, 128, tpw, tpw * 128, wgs, nbuf)
    run(f"ring {nbuf} x 32 KiB, {per_cu} WG/CU | rows 1.5 MB apart, tiles interleaved over WGs  ", B * 65536, 128 * wgs, tpw, 128, wgs, nbuf)
    run(f"ring {nbuf} x 32 KiB, {per_cu} WG/CU | tile-major layout (32 KiB contiguous per tile)  ", 128, 32768, tpw, tpw * 32768, wgs, nbuf)
tpw = B * 512 // 256
run("ring 4 x 32 KiB, 1 WG/CU | rows 1.5 MB apart, contiguous range, cache policy nt|sc1           ", B * 65536, 128, tpw, tpw * 128, 256, 418)
run("ring 4 x 32 KiB, 1 WG/CU | tile-major layout, cache policy nt|sc1                            ", 128, 32768, tpw, tpw * 32768, 256, 418)
for nmfma, nlds in ((8, 0), (16, 0), (20, 0), (32, 0), (0, 16), (0, 32), (20, 32)):
    run(f"ring 4, 8 waves, + {nmfma} MFMA 32x32x16 and {nlds} dependent tr-reads per wave per tile", B * 65536, 128, tpw, tpw * 128, 256, 4, nmfma, nlds)
