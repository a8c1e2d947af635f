"""bench.py's contract with the driver: ONE JSON line with the agreed keys, the roofline and cpu_baseline objects, exactly the
requested number of timed steps (a short run of the default workload, as a child process like the driver starts it)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--min-seconds", "0.05",
                          "--cpu-sample-rays", "64", "--secondary", "", "--trained-steps", "40", "--full-model", "mixed8192_vod_nll", "--full-model-trained-steps", "40"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert base["metric"].startswith(r["metric"]) and r["unit"] == "rays/s"  # (BASELINE's metric string goes on with PSNR / GPU counts)
    assert r["n_gpus"] == 1 and r["steps"] == 6 and r["warmup"] == 2
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["data"] == "synthetic" and r["dtype"] in ("bf16", "f16", "f32")
    assert r["config"]["workload"] == "mixed16384_neuradar" and "model" not in r["config"]
    assert r["value"] > 0 and abs(r["value"] - 16384 / (r["ms_per_step"] * 1e-3)) < 0.01 * r["value"]
    roof = r["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["traffic"] is None or roof["traffic"] > 0
    assert abs(roof["achieved"] - roof["bytes_per_launch"] / (roof["avg_us"] * 1e-6) / 1e9) < 0.01 * roof["achieved"]
    # round 3: the dominant site by itself, the MFMA counter of the field kernels, the trained regime, the decoder workloads
    assert roof["serialised_us"] > 0 and roof["serialised_us"] <= roof["avg_us"] * 1.25
    assert abs(roof["frac_serialised"] - roof["bytes_per_launch"] / (roof["serialised_us"] * 1e-6) / 1e9 / roof["peak"]) < 1e-3
    # round 6: `frac` comes from the SERIALISED bracket (the site's kernels alone on the chip: reproducible, what rocprofv3's kernel
    # averages add up to); the in-step bracket of the site that stays in flight longest is kept under `in_step`; `traffic` names
    # the committed profile it was read from
    assert roof["timing"].startswith("SERIALISED") and roof["serialised_us"] == roof["avg_us"] and roof["frac_serialised"] == roof["frac"]
    ins = roof["in_step"]
    assert ins["timing"].startswith("IN-STEP") and ins["avg_us"] >= 0.8 * ins["serialised_us"] and abs(ins["frac"] - ins["achieved"] / 8000.0) < 1e-3
    assert (roof["traffic"] is None) == (roof["traffic_source"] is None) and (roof["traffic"] is None or "NOT measured in this run" in roof["traffic_source"])
    assert roof["mfma_busy_frac"] is None or all(0 <= v <= 1 for k, v in roof["mfma_busy_frac"].items() if k != "source")
    tr = r["trained"]
    assert tr["steps_trained"] == 40 and tr["value"] > 0 and tr["roofline"]["bound"] == "hbm" and tr["roofline"]["serialised_us"] > 0
    fm = r["full_model"]
    assert len(fm) == 1 and fm[0]["workload"] == "mixed8192_vod_nll" and fm[0]["value"] > 0 and fm[0]["decoders_us_in_step"] > 0
    assert fm[0]["rays"] == {"camera": 2048, "lidar": 1599, "radar": 4545} and fm[0]["radar_loss"] == "nll"
    # the rendering entry with the model the step trained: one camera image at a third of the resolution, one radar scan
    assert fm[0]["after_training"]["steps_trained"] == 40 and fm[0]["after_training"]["value"] > 0
    rd = fm[0]["render"]
    assert rd["camera_image"]["rays"] == 640 * 360 and rd["camera_image"]["outputs"]["rgb"] == [1080, 1920, 3] and rd["camera_image"]["ms"] > 0
    assert rd["radar_scan"]["rays"] == 4545 and rd["radar_scan"]["outputs"]["radar_output"] == [1, 4545, 7]
    cpu = r["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["unit"] == "rays/s" and cpu["sample"]
    # round 4: what limits the dominant site in words, the effective operand type, the scatters' arithmetic, the convergence
    # figures beside the speed (BASELINE's metric is "rays/sec + PSNR"), the decoder workloads' step times in `config`
    assert "limiter" in roof and "LDS" in roof["limiter"]
    assert r["config"]["mlp_operands"] == "bfloat16" and "32-bit fixed point" in r["config"]["tables_and_accumulation"]
    assert r["config"]["full_model_ms_per_step"]["mixed8192_vod_nll"]["fresh"] == fm[0]["ms_per_step"]
    assert r["config"]["full_model_ms_per_step"]["mixed8192_vod_nll"]["trained"] == fm[0]["after_training"]["ms_per_step"]
    q = tr["quality"]
    assert q["feature_psnr_db"] > 0 and q["depth_l1_m"] >= 0 and q["depth_l1_m_lidar_rays"] >= 0
    qf = fm[0]["after_training"]["quality"]
    assert qf["image_psnr_db"] > 0 and qf["depth_l1_m_lidar_rays"] >= 0
