"""The driver keeps an 8 KB tail of bench.py's stdout and parses its LAST line (a 28.7 KB line left `parsed: null`).
bench.compact_line() must turn any full record into one line < 4 KB that carries every field of the bench contract."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (torch is imported lazily inside main())

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _stub(name):
    return json.load(open(os.path.join(ROOT, "profiles", name)))


@pytest.mark.parametrize("name", ["r04_bench_line_20steps.json", "r04_bench_line.json", "r03_bench_line.json"])
def test_compact_line_from_a_committed_full_record(name):
    full = _stub(name)
    assert len(json.dumps(full)) > 6000          # the stub really is one of the oversized lines
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert len(line) < bench.COMPACT_LIMIT and "\n" not in line
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert d["value"] == pytest.approx(full["value"], rel=1e-6) and d["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert d["config"]["workload"] and "model" not in d["config"]
    ro = d["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s") and ro["peak"] == 8000.0
    assert ro["frac"] == pytest.approx(ro["achieved"] / ro["peak"], rel=1e-3)
    assert d["cpu_baseline"]["kind"] in ("reference", "port", "aten-restatement") and d["cpu_baseline"]["cores"] >= 1
    assert set(d["extra"]) == {"C2", "C3", "C4"} and all(v["train"] > 0 and v["eval"] > 0 for v in d["extra"].values())
    assert d["eval"]["value"] > 0 and d["eval"]["roofline"]["frac"] > 0


def test_the_drivers_view_of_stdout():
    """What the driver does: keep the last 8 KB of stdout, parse the last line -- with the complete record printed first (--full-line)."""
    full = _stub("r04_bench_line_20steps.json")
    code = ("import json,sys; sys.path.insert(0, %r); import bench; full=json.load(open(%r)); print('noise'); print(json.dumps(full)); "
            "print(bench.compact_line(full, None))" % (ROOT, os.path.join(ROOT, "profiles", "r04_bench_line_20steps.json")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout
    assert len(out) > 20000
    d = json.loads(out[-8192:].splitlines()[-1])
    assert d["value"] == pytest.approx(full["value"], rel=1e-6) and d["n_gpus"] == 1


def test_compact_line_sheds_optional_parts_rather_than_overflow():
    full = _stub("r04_bench_line_20steps.json")
    full["config"]["workload"] = "w" * 5000
    full["extra"] = {"C%d" % i: dict(full["extra"]["C2"]) for i in range(40)}
    line = bench.compact_line(full, None)
    assert len(line) < bench.COMPACT_LIMIT
    d = json.loads(line)
    assert "extra" not in d and d["value"] > 0 and "roofline" in d


def test_multi_gpu_fields_ride_in_the_compact_line():
    full = _stub("r04_bench_line_20steps.json")
    full.pop("cpu_baseline")
    full.pop("extra")
    full.update(n_gpus=8, phases_us={"compute": 33.0, "reduce_scatter": 51.0, "optimiser": 8.0, "all_gather": 0.0, "row_norms": 5.0, "steps": 16},
                predicted_step_us=bench.predicted_step_us(8, True), replicas_identical=True,
                collectives={"backend": "nccl", "world_size": 8, "per_step": "all_reduce(flat grad, 6518400 B)", "captured": False})
    d = json.loads(bench.compact_line(full, None))
    assert d["phases_us"]["compute"] == 33.0 and d["replicas_identical"] is True and d["collectives"]["world_size"] == 8
    assert d["predicted_step_us"]["step"] == pytest.approx(33 + 25 + 2 * 6518400 / 8 / 61e3 + 8 + 5, rel=1e-3)


def test_rccl_setup_summary_reads_an_init_log(tmp_path, monkeypatch):
    """bench.rccl_setup_summary on the shape of log RCCL writes at communicator creation (NCCL_DEBUG=INFO, INIT [+ TUNING]); a
    missing or unrecognisable log gives None, never an exception."""
    import os
    log = "/tmp/kge_rccl_%d.log" % os.getpid()
    if os.path.exists(log):
        os.remove(log)
    assert bench.rccl_setup_summary() is None
    with open(log, "w") as f:
        f.write("box:1:1 [0] NCCL INFO RCCL version 2.22.3+hip7.0 HEAD:abc\n"
                "box:1:2 [0] NCCL INFO Channel 00/16 :    0   1   2   3   4   5   6   7\n"
                "box:1:2 [0] NCCL INFO Channel 15/16 :    0   7   6   5   4   3   2   1\n"
                "box:1:2 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC\n"
                "box:1:2 [0] NCCL INFO Connected all rings\nbox:1:2 [0] NCCL INFO Connected all trees\n"
                "box:1:2 [0] NCCL INFO 16 coll channels, 0 collnet channels, 0 nvls channels, 16 p2p channels, 2 p2p channels per peer\n"
                "box:1:2 [0] NCCL INFO AllReduce: 6518400 Bytes -> Algo 1 proto 2 time 45.1\n")
    try:
        got = bench.rccl_setup_summary()
    finally:
        os.remove(log)
    assert got["version"].startswith("2.22.3") and got["channels"] == 16 and got["transports"] == ["P2P/IPC"]
    assert got["connected"] == ["rings", "trees"] and got["chosen"] == {"AllReduce 6518400 B": "ring/simple"}
    with open(log, "w") as f:
        f.write("nothing recognisable\n")
    try:
        assert bench.rccl_setup_summary() is None
    finally:
        os.remove(log)
