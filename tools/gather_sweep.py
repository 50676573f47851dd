"""In-situ sweep of the gather kernel's tiling (the real persona step: gather then a prefill that reads the rows)."""
import os, subprocess, sys, json
combos = [(65536, 2, 16), (65536, 1, 16), (131072, 1, 16), (32768, 2, 16), (65536, 2, 8), (65536, 4, 16), (131072, 2, 16), (32768, 4, 16), (65536, 3, 16)]
for tile, ppw, unroll in combos:
    env = dict(os.environ, PC_GATHER_TILE=str(tile), PC_GATHER_PPW=str(ppw), PC_GATHER_UNROLL=str(unroll))
    out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"], env=env,
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print(f"tile={tile:6d} ppw={ppw:2d} unroll={unroll:2d}: gather {d['roofline']['avg_launch_us']:.1f} us "
          f"{d['roofline']['achieved']:.0f} GB/s  ttft {d['ttft_ms']:.3f} ms", flush=True)
