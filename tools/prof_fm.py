"""Runs nn.SpatialMatching on K-plane feature maps N times for profiling (rocprofv3 kernel stats / PMC passes):
    python tools/prof_fm.py <shape> [n]
shape: v2      = version2/network.lua:30 at VGA: SpatialMatching(17,17) on 32 planes (in1 448x608 after the crop)
       k32     = SpatialMatching(16,16), K = 32, 625x465 (VGA minus the window)
       k10     = SpatialMatching(16,16), K = 10, 625x465
       tm      = tests/time_matching.lua:18: SpatialMatching(16,16), K = 10, 320x180 minus the three 5x5 layers and the window
Prints the HIP-event time per call, the output bytes and the 2K-lane-op VALU floor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import depth_estimation_amd as d

SHAPES = {
    "v2": (32, 480 - 16 - 16, 640 - 16 - 16, 17),
    "k32": (32, 480 - 15, 640 - 15, 16),
    "k10": (10, 480 - 15, 640 - 15, 16),
    "tm": (10, 180 - 12 - 15, 320 - 12 - 15, 16),
    "k32-720p": (32, 720 - 15, 1280 - 15, 16),
    "k32-r4": (32, 512, 512, 16),     # 1024 tiles of the flat kernel: exactly 4 rounds on 256 CUs
    "k32-r45": (32, 576, 512, 16),    # 1152 tiles: 4.5 rounds
    "k32-r1": (32, 128, 512, 16),     # 256 tiles: one round
    "k32-r05": (32, 64, 512, 16),     # 128 tiles: half of the CUs
}


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "k32"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    K, H1, W1, win = SHAPES[shape]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    in1 = torch.rand((K, H1, W1), generator=g).to(dev)
    in2 = torch.rand((K, H1 + win - 1, W1 + win - 1), generator=g).to(dev)
    m = d.nn.SpatialMatching(win, win, False)
    for _ in range(3):
        out = m.forward([in1, in2])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        out = m.forward([in1, in2])
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    byts = out.numel() * 4 + in1.numel() * 4 + in2.numel() * 4
    laneops = 3.0 * out.numel() * K
    print("%s: K=%d %dx%d win %d: %.4f ms | %.1f MB -> %.2f TB/s (%.3f of 8) | %.2e lane-ops (%.1f us at 2.5 cyc/inst/SIMD, 2.4 GHz) | kernel %s"
          % (shape, K, W1, H1, win, ms, byts / 1e6, byts / ms / 1e9, byts / ms / 1e9 / 8.0, laneops, laneops / 64 / 1024 * 2.5 / 2.4e3, d.get_ctx(0).last_kernel()))


if __name__ == "__main__":
    main()
