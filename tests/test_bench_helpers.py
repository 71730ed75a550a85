"""bench.py's host-side bookkeeping (no GPU): defaults of the driver contract and the per-launch roofline aggregation."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)              # `main()` only runs under __main__
    return mod


class _Event:
    def __init__(self, t_ms):
        self.t = t_ms

    def elapsed_time(self, other):
        return other.t - self.t


def test_defaults_follow_the_driver_contract(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.workload, a.precision) == (1, "c3", b.DEFAULT_PRECISION) and a.steps >= 5 and a.warmup >= 1
    assert a.precision != "bf16" and a.precision in b.DTYPE_NAMES      # the headline is a reference-precision number
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)


def test_conv_timer_aggregates_launch_records():
    b = _bench()
    t = b.ConvTimer()
    t.records = [("conv_igemm", 2e12, _Event(0), _Event(2.0), "fwd 128->128 in 16x256x256 k3 s1 up1"),
                 ("conv_igemm", 2e12, _Event(0), _Event(2.0), "fwd 128->128 in 16x256x256 k3 s1 up1"),
                 ("conv_igemm", 1e12, _Event(0), _Event(1.0), "dgrad 3->64 in 16x256x256 k3 s1 up1"),
                 ("conv_wgrad", 4e12, _Event(0), _Event(2.0), "wgrad 512->512 in 16x64x64 k3 s1 up2sub"),
                 ("conv_igemm", 1e12, _Event(0), _Event(5.0), "fwd 256->128 in 16x256x256 k1 s1 up1")]
    s = t.summary()
    assert s["conv_igemm"][0] == 4 and abs(s["conv_igemm"][1] - 6e12) < 1 and abs(s["conv_igemm"][2] - 0.010) < 1e-9
    assert s["conv_wgrad"][0] == 1
    # the 3x3 GEMMs with >= 64 channels on both sides, all passes: two forwards + the sub-pixel weight gradient
    n, fl, sec = t.family(lambda what, ci, co, k, st: k == 3 and ci >= 64 and co >= 64)
    assert n == 3 and abs(fl - 8e12) < 1 and abs(sec - 0.006) < 1e-9
    lines = t.table(steps=1).splitlines()
    assert lines[0].split()[0] == "layer" and lines[1].startswith("fwd 256->128")           # sorted by time
    row = [ln for ln in lines if ln.startswith("fwd 128->128")][0].split()
    assert float(row[-4]) == 2.0 and abs(float(row[-1]) - 1000.0) < 1e-6                     # 2 launches / step, 1000 TFLOP/s


def test_hbm_family_rows():
    b = _bench()
    t = b.ConvTimer()
    t.records = [("hbm:gn_apply", 4e9, _Event(0), _Event(1.0), ""), ("hbm:gn_apply", 4e9, _Event(0), _Event(1.0), ""),
                 ("hbm:adamw", 2.8e9, _Event(0), _Event(0.5), ""), ("conv_igemm", 1e12, _Event(0), _Event(1.0), "x")]
    rows = t.hbm_rows(steps=2)
    assert [r["kernel"] for r in rows] == ["gn_apply", "adamw"]
    assert rows[0]["GB/s"] == 4000.0 and rows[0]["frac_of_8TBps"] == 0.5 and rows[0]["calls_per_step"] == 1.0
    assert rows[1]["GB/s"] == 5600.0 and rows[1]["ms_per_step"] == 0.25


def test_cpu_baseline_and_parity_legs_on_the_emulator(backend):
    """The two legs bench.py runs after the timed region — the oracle's timed steps and the HIP step on the oracle's
    batch and weights — on a toy model (the emulator stands in for the GPU; `backend` also runs it on the MI355X)."""
    import argparse
    b = _bench()
    cfg = {"ch": 32, "ch_mult": (1, 2), "z": 4, "res": 16, "gan": backend.name == "gpu", "vq": None}
    line, ref = b.cpu_baseline(argparse.Namespace(cpu_baseline_res=16), cfg, configs0=False)
    assert line["unit"] == "images/sec" and line["value"] > 0 and line["kind"] == "port" and "median" in line["sample"]
    par = b.parity_vs_oracle("fp32x3", ref, backend.device)
    assert par["perceptual_loss_rel"] < 1e-4 and par["overall_vae_loss_rel"] < 1e-4 and par["recon_rel"] < 2e-4, par
    if backend.name == "gpu":
        assert par["d_loss_rel"] < 1e-4
        par = b.parity_vs_oracle("bf16", ref, backend.device)
        assert par["perceptual_loss_rel"] < 5e-2 and par["recon_rel"] < 5e-2, par


def test_traffic_file_is_only_trusted_for_these_sources_this_precision_and_this_workload(tmp_path, monkeypatch):
    """`roofline.traffic` comes from a committed PMC pass (profiles/r*_traffic.json, tools/gpu_traffic.sh).  bench.load_traffic only
    uses a file measured on THESE kernel sources (`sources_sha`), at the timed precision and on the timed workload; anything else
    prints `traffic: null` with the reason — never stale or foreign bytes."""
    import json
    b = _bench()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    assert b.load_traffic("ref") == (None, None, "no profiles/r*_traffic.json")
    # the sources hash is computed under ROOT: give it the real tree's hash through a stub
    monkeypatch.setattr(b, "kernel_sources_sha", lambda: "feedbeef00000000")
    stale = {"igemm_family_bytes_per_launch": 1, "precision": "ref", "sources_sha": "0123456789abcdef"}
    (tmp_path / "profiles" / "r1_traffic.json").write_text(json.dumps(stale))
    info, src, why = b.load_traffic("ref")
    assert info is None and src is None and "other kernel sources" in why
    good = dict(stale, sources_sha="feedbeef00000000", igemm_family_bytes_per_launch=214)
    (tmp_path / "profiles" / "r2_traffic.json").write_text(json.dumps(good))
    info, src, why = b.load_traffic("ref")
    assert info["igemm_family_bytes_per_launch"] == 214 and src.endswith("r2_traffic.json") and why is None
    assert b.load_traffic("bf16")[0] is None and "precision" in b.load_traffic("bf16")[2]
    info, src, why = b.load_traffic("ref", "c5")          # files without a `workload` key were measured on the default one
    assert info is None and "workload c3, not c5" in why
    (tmp_path / "profiles" / "r3_traffic.json").write_text(json.dumps(dict(good, workload="c5", igemm_family_bytes_per_launch=999)))
    assert b.load_traffic("ref", "c5")[0]["igemm_family_bytes_per_launch"] == 999
    assert b.load_traffic("ref", "c3")[0]["igemm_family_bytes_per_launch"] == 214
