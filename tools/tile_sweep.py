import sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import conv_bench
from score_sde_pytorch_amd import _lib as L
for (cin, cout, h) in [(256, 256, 4), (512, 256, 4)]:
    for gn in (0, 1):
        for tile, name in [(L.TILE_AUTO, "auto"), (L.TILE_256x64, "256x64"), (L.TILE_128x64, "128x64"), (L.TILE_64x64, "64x64"), (L.TILE_256x32, "256x32")]:
            t, ms = conv_bench.time_conv(256, cin, cout, h, tile, gn, reps=10)
            print("B=256 %d->%d @%dx%d gn=%d tile %-7s %6.1f TF/s (%.3f ms)" % (cin, cout, h, h, gn, name, t, ms), flush=True)
