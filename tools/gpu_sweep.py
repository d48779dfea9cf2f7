"""Throughput sweeps on the GPU (tuning aid, not part of the product)."""
import os, sys, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def run(env, packets=2e7, extra=()):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--packets", str(packets), "--no-cpu-baseline", *extra],
                         capture_output=True, text=True, env=e, timeout=150)
    try:
        j = json.loads(out.stdout.strip().split("\n")[-1])
        return "%.3e pk/s  kernel %.1f ms  %.1f cross/pk  %.3e cross/s" % (
            j["value"], j["roofline"]["kernel_ms"], j["config"]["crossings_per_packet"],
            j["value"] * j["config"]["crossings_per_packet"])
    except Exception:
        return out.stdout[-300:] + out.stderr[-600:]
if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "mode"):
        print("hbm default", run({"MCGPU_DEPOSIT": "hbm"}), flush=True)
        print("lds default", run({"MCGPU_DEPOSIT": "lds"}), flush=True)
        print("lds nodeposit", run({"MCGPU_DEPOSIT": "lds", "MCGPU_DIAG_FLAGS": "1"}), flush=True)
    if which in ("all", "inner"):
        for it in (8, 16, 32, 64, 128, 256):
            print("lds inner_iters", it, run({"MCGPU_INNER_ITERS": str(it)}), flush=True)
    if which in ("all", "flush"):
        for f in (1, 4, 16, 64, 256):
            print("lds flush_every", f, run({"MCGPU_FLUSH_EVERY": str(f)}), flush=True)
    if which in ("all", "block"):
        for bt in (256, 384, 512):
            print("lds block_threads", bt, run({}, extra=("--block-threads", str(bt))), flush=True)
    if which in ("all", "big"):
        print("lds 1e8", run({}, packets=1e8), flush=True)
        print("pascucci 2e7", run({}, extra=("--config", "pascucci")), flush=True)
        print("3d 5e6", run({}, packets=5e6, extra=("--config", "ref41_3d")), flush=True)
