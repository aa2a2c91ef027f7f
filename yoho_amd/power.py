"""Clock / power sampling beside a measurement (bench.py, tools/archive/sweep_partI_chunk.py).

The irrep GEMMs of PartI are limited by the package power budget, not by their schedule: the evidence so far was
GRBM_GUI_ACTIVE / duration from PMC runs.  This module gives the two independent readings the bench line carries:

  * PowerMonitor - a side thread polling the SMU through amdsmi (shader clock per XCD, socket power, power cap) while the
    timed region runs; sysfs (hwmon / pp_dpm_sclk) if amdsmi cannot be initialised;
  * ClockProbe   - the library's one-wave probe kernel (include/yoho_hip.h, yoho_clock_probe) on a high-priority stream of its
    own: shader cycles per constant-rate wall tick while the kernels under test run on the other streams.

Both degrade to None fields instead of failing: a box without SMU access must still produce the bench line.
"""
import glob
import os
import threading
import time


def _mean(v):
    v = [x for x in v if x is not None]
    return (sum(v) / len(v)) if v else None


class PowerMonitor:
    def __init__(self, device_index=0, period_s=0.002):
        self.dev = int(device_index)
        self.period = period_s
        self.samples = []            # (t, sclk_mhz, power_w)
        self.cap_w = None
        self.source = None
        self._stop = threading.Event()
        self._thr = None
        self._h = None
        self._smi = None
        self._hwmon = None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self._h = hs[self.dev if self.dev < len(hs) else 0]
            self._smi = amdsmi
            self.source = "amdsmi"
            try:
                cap = amdsmi.amdsmi_get_power_cap_info(self._h)["power_cap"]
                self.cap_w = cap / 1e6 if cap and cap > 10000 else cap          # microwatts in this amdsmi, watts in older ones
            except Exception:
                pass
        except Exception:
            self._smi = None
            cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
            if cands:
                self._hwmon = cands[self.dev if self.dev < len(cands) else 0]
                self.source = "sysfs"
                try:
                    self.cap_w = int(open(os.path.join(self._hwmon, "power1_cap")).read()) / 1e6
                except Exception:
                    pass

    @property
    def available(self):
        return self.source is not None

    def _num(self, x):
        return x if isinstance(x, (int, float)) and not isinstance(x, bool) else None

    def _read(self):
        if self._smi is not None:
            s = self._smi
            try:
                m = s.amdsmi_get_gpu_metrics_info(self._h)
                clks = [self._num(c) for c in (m.get("current_gfxclks") or [])]
                clks = [c for c in clks if c and c < 10000]
                sclk = _mean(clks) if clks else self._num(m.get("current_gfxclk"))
                pw = self._num(m.get("current_socket_power"))
                if pw is None:
                    pw = self._num(m.get("average_socket_power"))
                if sclk is not None or pw is not None:
                    return sclk, pw
            except Exception:
                pass
            sclk = pw = None
            try:
                sclk = self._num(s.amdsmi_get_clock_info(self._h, s.AmdSmiClkType.GFX)["clk"])
            except Exception:
                pass
            try:
                p = s.amdsmi_get_power_info(self._h)
                pw = self._num(p.get("current_socket_power"))
                if pw is None:
                    pw = self._num(p.get("socket_power"))
                if pw is None:
                    pw = self._num(p.get("average_socket_power"))
            except Exception:
                pass
            return sclk, pw
        if self._hwmon is not None:
            sclk = pw = None
            try:
                sclk = int(open(os.path.join(self._hwmon, "freq1_input")).read()) / 1e6
            except Exception:
                pass
            for f in ("power1_input", "power1_average"):
                try:
                    pw = int(open(os.path.join(self._hwmon, f)).read()) / 1e6
                    break
                except Exception:
                    pass
            return sclk, pw
        return None, None

    def read_once(self):
        """one SMU reading -> {"sclk_mhz", "power_w"} (None fields without SMU access)"""
        sclk, pw = self._read() if self.available else (None, None)
        return {"sclk_mhz": None if sclk is None else round(float(sclk), 1), "power_w": None if pw is None else round(float(pw), 1)}

    def _loop(self):
        while not self._stop.is_set():
            sclk, pw = self._read()
            self.samples.append((time.perf_counter(), sclk, pw))
            self._stop.wait(self.period)

    def start(self):
        self.samples = []
        if not self.available:
            return self
        self._stop.clear()
        self._thr = threading.Thread(target=self._loop, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        """-> {"sclk_mhz_mean", "sclk_mhz_min", "sclk_mhz_max", "power_w_mean", "power_w_max", "power_cap_w", "samples", "source"}"""
        if self._thr is not None:
            self._stop.set()
            self._thr.join()
            self._thr = None
        ck = [s[1] for s in self.samples if s[1] is not None]
        pw = [s[2] for s in self.samples if s[2] is not None]
        r = lambda x: None if x is None else round(float(x), 1)
        return {"sclk_mhz_mean": r(_mean(ck)), "sclk_mhz_min": r(min(ck) if ck else None), "sclk_mhz_max": r(max(ck) if ck else None),
                "power_w_mean": r(_mean(pw)), "power_w_max": r(max(pw) if pw else None), "power_cap_w": r(self.cap_w),
                "samples": len(self.samples), "source": self.source}

    def __enter__(self):
        return self.start()

    def __exit__(self, *a):
        self.result = self.stop()
        return False


class ClockProbe:
    """Shader clock under load from the device itself: probes of `us` microseconds queued back to back on a high-priority stream
    while the caller's work runs on other streams.  mhz() waits for them and returns the per-probe clocks."""

    def __init__(self, ctx, us=200, capacity=4096):
        import torch
        self.ctx, self.us = ctx, int(us)
        self.stream = torch.cuda.Stream(priority=-1)
        # one result buffer, zeroed once and synchronised: a per-probe torch.zeros would put a fill kernel on the CALLER's stream,
        # behind the kernels under test, and wipe the probe's answer when it finally runs
        # (zeros made on the host and copied: no tensor-library fill kernel - bench.py --timed-only shows a trace without any)
        self.buf = torch.zeros((capacity, 3), dtype=torch.int64).to(f"cuda:{ctx.device}")
        torch.cuda.synchronize()
        self.n = 0
        self.out = []

    def queue(self, n=1):
        for _ in range(n):
            if self.n >= self.buf.shape[0]:
                return
            row = self.buf[self.n]
            self.ctx.clock_probe(self.us, self.stream, out=row)
            self.out.append(row)
            self.n += 1

    def mhz(self):
        import torch
        self.stream.synchronize()
        vals = []
        for c, w, khz in self.buf[:self.n].cpu().tolist():
            if w > 0:
                vals.append(c / w * khz / 1000.0)
        self.out = []
        self.n = 0
        self.buf.zero_()
        torch.cuda.synchronize()
        return vals

    def summary(self):
        v = self.mhz()
        if not v:
            return None
        return {"shader_mhz_mean": round(sum(v) / len(v), 1), "shader_mhz_min": round(min(v), 1), "shader_mhz_max": round(max(v), 1),
                "probes": len(v), "probe_us": self.us}
