"""Throughput sweeps on the GPU (tuning aid, not part of the product)."""
import os, sys, time, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
def run(env, packets=2e7, extra=()):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--packets", str(packets), "--no-cpu-baseline", *extra],
                         capture_output=True, text=True, env=e)
    try:
        j = json.loads(out.stdout.strip().split("\n")[-1])
        return j["value"], j["roofline"]["kernel_ms"]
    except Exception:
        return out.stdout[-300:] + out.stderr[-300:], None
if __name__ == "__main__":
    for it in (2, 4, 8, 16, 32, 64):
        print("inner_iters", it, run({"MCGPU_INNER_ITERS": str(it)}), flush=True)
    print("no deposit", run({"MCGPU_DIAG_FLAGS": "1"}), flush=True)
    for gb in (256, 512, 768, 1024, 2048):
        print("grid_blocks", gb, run({}, extra=("--grid-blocks", str(gb))), flush=True)
    for bt in (64, 128, 256):
        print("block_threads", bt, run({}, extra=("--block-threads", str(bt))), flush=True)
    print("1e8", run({}, packets=1e8), flush=True)
