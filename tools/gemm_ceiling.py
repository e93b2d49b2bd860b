#!/usr/bin/env python
"""What a plain library GEMM reaches on this board under its power cap (torch.matmul -> hipBLASLt), as a yardstick for the
MFMA fractions bench.py reports: the 2.5 PFLOP/s dense 16-bit peak assumes 2.4 GHz, the board holds 1.4 kW.  (GPU box)"""
import subprocess
import threading
import time

import torch


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if "Power" in l or "sclk" in l]
        return " | ".join(keep)
    except Exception as e:      # noqa: BLE001
        return f"(rocm-smi: {e})"


def main():
    for dt in (torch.bfloat16, torch.float16):
        for n in (4096, 8192, 16384):
            a = torch.randn(n, n, device="cuda", dtype=dt)
            b = torch.randn(n, n, device="cuda", dtype=dt)
            for _ in range(3):
                a @ b
            torch.cuda.synchronize()
            iters = max(10, int(2.0e15 / (2.0 * n ** 3)))          # ~1-2 s of work: long enough for the power cap to bite
            probe = {}
            th = threading.Timer(0.8, lambda: probe.setdefault("smi", smi()))
            th.start()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                a @ b
            e1.record()
            torch.cuda.synchronize()
            th.join()
            ms = e0.elapsed_time(e1) / iters
            print(f"{str(dt):16s} {n:6d}^3: {ms:8.3f} ms  {2.0 * n ** 3 / ms / 1e9:8.1f} TFLOP/s   {probe.get('smi', '')}", flush=True)


if __name__ == "__main__":
    main()
