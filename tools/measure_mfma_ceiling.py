"""Ceiling of the matrix pipe under the socket power cap: afk_mfma_ceiling mode 0 (register-resident N(0,1) operands, no memory
access in the loop) and mode 1 (+ the 256x256 GEMM's LDS fragment reads), ~2 s each on all 256 CUs x 8 waves, with clock and socket power sampled
from sysfs / rocm-smi beside it; then the GEMM itself on the gate|up shape on N(0,1) operands and on zeros.
    python tools/measure_mfma_ceiling.py [out.json]        -> profiles/r06_mfma_power_ceiling.md is written from this"""
import ctypes
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from audio_flamingo_amd import _lib, ops


def _read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


class Sampler(threading.Thread):
    """socket power (hwmon power1_average / power1_input, microwatts) and the current sclk level while a leg runs"""

    def __init__(self):
        super().__init__(daemon=True)
        self.stop_, self.power, self.sclk = False, [], []
        self.pw = [p for pat in ("power1_average", "power1_input") for p in glob.glob(f"/sys/class/drm/card*/device/hwmon/hwmon*/{pat}")]
        self.clk = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
        self.freq = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")

    def run(self):
        while not self.stop_:
            for p in self.pw[:1]:
                v = _read(p)
                if v and v.isdigit():
                    self.power.append(int(v) / 1e6)
            for p in self.freq[:1]:
                v = _read(p)
                if v and v.isdigit():
                    self.sclk.append(int(v) / 1e6)
            if not self.freq:
                for p in self.clk[:1]:
                    v = _read(p) or ""
                    for ln in v.splitlines():
                        if ln.endswith("*"):
                            try:
                                self.sclk.append(float(ln.split(":")[1].replace("Mhz", "").replace("*", "").strip()))
                            except ValueError:
                                pass
            time.sleep(0.02)

    def summary(self):
        def st(x):
            x = x[len(x) // 4:]   # the first quarter is the ramp
            return None if not x else {"mean": round(sum(x) / len(x), 1), "min": round(min(x), 1), "max": round(max(x), 1), "n": len(x)}
        return {"socket_power_w": st(self.power), "sclk_mhz": st(self.sclk)}


def timed(fn, seconds):
    """launch fn() repeatedly for ~seconds; -> (total ms by HIP events, launches, sampler summary)"""
    s = Sampler()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.start()
    e0.record()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            fn()
            n += 1
        torch.cuda.current_stream().synchronize() if n % 32 == 0 else None
    e1.record()
    torch.cuda.synchronize()
    s.stop_ = True
    s.join()
    return e0.elapsed_time(e1), n, s.summary()


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    seed = torch.randn(1 << 20, device=dev).to(torch.bfloat16)
    sink = torch.zeros(4, device=dev)
    out = {"afk_build_id": _lib.load().afk_build_id().decode(), "device": torch.cuda.get_device_name(0), "legs": {}}
    stream = torch.cuda.current_stream().cuda_stream
    for name, mode, operands in (("mfma_registers_n01", 0, seed), ("mfma_lds_reads_n01", 1, seed), ("mfma_registers_zeros", 0, torch.zeros_like(seed))):
        fl = ctypes.c_double(0)
        iters = 20000     # 256 blocks x 8 waves x 20000 x 16 MFMAs = 21.5 TFLOP per launch (~12 ms at 1.8 PF)

        def go():
            _lib.call("afk_mfma_ceiling", mode, 256, iters, operands.data_ptr(), sink.data_ptr(), ctypes.byref(fl), stream)
        ms, n, smp = timed(go, 2.0)
        out["legs"][name] = {"tflops": fl.value * n / (ms * 1e-3) / 1e12, "ms_per_launch": ms / n, "launches": n, **smp}
        print(name, json.dumps(out["legs"][name]), flush=True)
        time.sleep(1.0)
    a = torch.randn(8192, 3584, device=dev).to(torch.bfloat16)
    w = torch.randn(37888, 3584, device=dev).to(torch.bfloat16)
    c = torch.empty((8192, 37888), device=dev, dtype=torch.bfloat16)
    for name, A, W in (("gemm_gate_up_n01", a, w), ("gemm_gate_up_zeros", torch.zeros_like(a), torch.zeros_like(w))):
        ms, n, smp = timed(lambda: ops.gemm_nt(A, W, out=c), 2.0)
        out["legs"][name] = {"tflops": 2.0 * 8192 * 37888 * 3584 * n / (ms * 1e-3) / 1e12, "ms_per_launch": ms / n, "launches": n, **smp}
        print(name, json.dumps(out["legs"][name]), flush=True)
        time.sleep(1.0)
    L = out["legs"]
    out["gemm_frac_of_power_ceiling"] = L["gemm_gate_up_n01"]["tflops"] / L["mfma_registers_n01"]["tflops"]
    out["gemm_frac_of_lds_ceiling"] = L["gemm_gate_up_n01"]["tflops"] / L["mfma_lds_reads_n01"]["tflops"]
    try:
        out["rocm_smi"] = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=30).stdout[-1500:]
    except Exception as e:
        out["rocm_smi"] = repr(e)
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "mfma_power_ceiling.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "rocm_smi"}))


if __name__ == "__main__":
    main()
