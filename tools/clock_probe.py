"""Shader clock and socket power the chip holds under the ring GEMM, or (--bench) under the whole training step: polls rocm-smi
while a worker thread / a bench.py child keeps the GPU busy.  usage: python tools/clock_probe.py [--bench [bench.py args]]"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def poll(tag, secs):
    t_end = time.time() + secs
    out = []
    while time.time() < t_end:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True)
        try:
            d = json.loads(r.stdout)
            c = next(iter(d.values()))
            out.append({k: v for k, v in c.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()})
        except Exception as e:  # noqa
            out.append({"err": r.stdout[:200] + r.stderr[:200]})
        time.sleep(0.3)
    print(tag, json.dumps(out[:8]), flush=True)


def busy(fn, stop):
    import torch
    while not stop.is_set():
        for _ in range(20):
            fn()
        torch.cuda.synchronize()


def bench_mode(args):
    """bench.py as a child process (the parent never touches the GPU); samples start once the child has warmed up."""
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "120", "--warmup", "5", *args],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    time.sleep(25.0)   # import + build check + warmup + host-side setup
    while child.poll() is None:
        poll("step", 1.5)
    print(child.stdout.read().strip()[:400], flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--bench":
        poll("idle", 1.0)
        return bench_mode(sys.argv[2:])
    import torch
    from deepcoro_clip_amd import kernels as K
    global torch
    dev = "cuda"
    poll("idle", 1.0)
    for name, (M, N, Kd) in {"ring 50208x1536x384": (50208, 1536, 384), "ring 8192^3": (8192, 8192, 8192)}.items():
        a = (torch.rand(M, Kd, device=dev) * 2 - 1).bfloat16()
        b = (torch.rand(N, Kd, device=dev) * 2 - 1).bfloat16()
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: K.gemm(a, b, out=c)
        fn(); torch.cuda.synchronize()
        stop = threading.Event()
        th = threading.Thread(target=busy, args=(fn, stop)); th.start()
        time.sleep(0.5)
        poll(name, 3.0)
        stop.set(); th.join()
        # throughput on the side
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        print(f"{name}: {us:.1f} us, {2 * M * N * Kd / us / 1e6:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
