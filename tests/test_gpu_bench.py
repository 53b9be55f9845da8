"""GPU test of the benchmark's contract: `python bench.py` prints exactly one JSON line on stdout with the
fields the driver reads, whatever the libraries under it print."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags):
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, capture_output=True,
                          text=True, timeout=280)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [line for line in done.stdout.splitlines() if line.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_one_json_line_with_the_contract_fields():
    line = run_bench("--steps", "32", "--warmup", "16", "--width", "480", "--height", "272", "--cpu-seconds", "1",
                     "--soup-sizes", "100000")
    assert line["metric"] == "closest_hit_mrays_per_s" and line["unit"] == "Mrays/s"
    assert line["n_gpus"] == 1 and line["steps"] == 32 and line["warmup"] == 16
    assert line["higher_is_better"] is True and line["scaling"] == "strong" and line["vs_baseline"] is None
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and "workload" in line["config"]
    assert line["value"] > 0 and line["ms_per_step"] > 0
    # value is the rays of the timed steps over their time
    assert abs(line["value"] - line["config"]["rays_per_step"] / line["ms_per_step"] / 1e3) < 1e-6 * line["value"]
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0 < roof["frac"] < 1
    assert roof["traffic"] is None          # the committed counters are of the 1080p frame
    assert roof["copy_ceiling_gbs"] > 1000
    # achieved = algorithmic bytes per launch over the launch time measured in this run
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * roof["achieved"]
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "Mrays/s" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
    assert cpu["gather"]["unit"] == "gathers/s" and cpu["gather"]["value"] > 0
    assert line["photon_gathers_per_s"] > 0
    # the legs that ride along: the final-profile workload and the triangle soup (the lab's 5,890 triangles live in
    # the caches, so its algorithmic bytes per second are no physical bound and can pass the HBM peak)
    lab = line["caustic_lab"]
    assert lab["ms_per_step"] > 0 and lab["mrays_per_s"] > 0 and lab["gathers_per_s_in_frame"] > 0
    assert lab["photon_pass_ms"] > 0 and lab["photons_stored_per_pass"] > 0 and lab["roofline"]["frac"] > 0
    assert abs(lab["mrays_per_s"] - lab["rays_per_step"] / lab["ms_per_step"] / 1e3) < 1e-6 * lab["mrays_per_s"]
    (soup,) = line["soup"]
    assert soup["triangles"] == 100000 and soup["build_ms"] > 0 and soup["mrays_per_s"] > 0 and 0 < soup["frac"] < 1


def test_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher around it starts two ranks itself (gloo here: both share the
    one GPU of the test box) and prints the one line of a two-rank run."""
    env_free = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                           "--steps", "16", "--warmup", "16", "--width", "480", "--height", "272",
                           "--no-cpu-baseline", "--no-photon-bench"], cwd=ROOT, capture_output=True, text=True,
                          timeout=280, env=env_free)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [line for line in done.stdout.splitlines() if line.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and len(line["multi_gpu"]["rays_per_rank"]) == 2
    assert all(r > 0 for r in line["multi_gpu"]["rays_per_rank"])
    assert abs(sum(line["multi_gpu"]["rays_per_rank"]) / 16 - line["config"]["rays_per_step"]) < 1e-6 * line["config"]["rays_per_step"]

