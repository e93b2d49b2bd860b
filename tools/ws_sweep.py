import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from sweep_thresholds import forward_ms
from capf.lib import PLAN_NO_WS
for bb, H, W, batches in (("hrnet_48", 256, 256, [8, 16, 24, 32, 48, 64, 96, 128, 256]), ("hrnet_32", 256, 256, [8, 16, 32, 64, 128, 256]), ("cpn", 384, 288, [4, 8, 16, 32, 64, 128])):
    for b in batches:
        d, a = forward_ms(bb, "bf16", b, 0, H, W, reps=12), forward_ms(bb, "bf16", b, PLAN_NO_WS, H, W, reps=12)
        print(f"{bb} bf16 batch {b:4d}: ws {d:8.3f} | no ws {a:8.3f}   {a / d:5.2f}x", flush=True)
