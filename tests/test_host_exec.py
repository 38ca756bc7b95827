"""The C++ host mirror (elaina-exec): host-only self test on CPU, and on the GPU box the whole
JSON-driven path (OBJ + colour file + conf.json -> run_expr -> raw field) against the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _exe():
    from elaina_amd import build
    return build.build_host()


def test_host_selftest():
    out = subprocess.run([_exe(), "--selftest"], capture_output=True, text=True)
    assert out.returncode == 0 and "selftest ok" in out.stdout, out.stdout + out.stderr


def test_missing_config_and_usage():
    exe = _exe()
    assert subprocess.run([exe], capture_output=True).returncode == 1
    out = subprocess.run([exe, "/nonexistent/conf.json"], capture_output=True, text=True)
    assert "does not exist" in out.stderr


def test_run_expr_without_gpu_fails_loudly(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import export_scene
    conf = export_scene.export("ladybug", str(tmp_path), frame=16, spp=1, depth=4)
    out = subprocess.run([_exe(), conf], capture_output=True, text=True)
    assert out.returncode == 1 and "no HIP device" in out.stderr
    assert os.path.exists(tmp_path / "exp" / "ladybug_u" / "conf.json")   # directory is created


@pytest.mark.gpu
def test_run_expr_end_to_end_matches_oracle(tmp_path, oracle, ladybug):
    import export_scene
    conf = export_scene.export("ladybug", str(tmp_path), frame=64, spp=8, depth=32)
    out = subprocess.run([_exe(), conf], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    exp = tmp_path / "exp" / "ladybug_u"
    res = json.load(open(exp / "result.json"))
    ref = oracle.solve(ladybug.as_dict(), 64, 64, 8, 32, 1.0, threads=os.cpu_count())
    assert res["walk_steps"] == ref["walk_steps"] and "duration" in res and "timestamp" in res
    field = export_scene.read_pfm(exp / "solution.pfm")
    assert np.array_equal(field, ref["field"])
    sdf = export_scene.read_pfm(exp / "dirichlet_sdf.pfm")[:, 0]
    assert np.array_equal(sdf, oracle.render_dirichlet_sdf(ladybug.as_dict(), 64, 64))
    assert os.path.exists(exp / "solution.ppm") and os.path.exists(exp / "solution_energy.pfm")


@pytest.mark.gpu
def test_run_expr_guided_configuration(tmp_path, ladybug):
    """the reference's n.json shape (type "guided" + network section) through the C++ host"""
    import export_scene
    from elaina_amd import UniformIntegrator, UniformIntegratorSettings
    conf = export_scene.export("ladybug", str(tmp_path), frame=192, spp=8, depth=48, integrator="guided")
    out = subprocess.run([_exe(), conf], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    exp = tmp_path / "exp" / "ladybug_n"
    res = json.load(open(exp / "result.json"))
    assert res["guided_steps"] > 0 and res["optimizer_steps"] > 0 and res["walk_steps"] > res["guided_steps"]
    assert "selection probability" in out.stderr          # print_network -> queryNetwork
    field = export_scene.read_pfm(exp / "solution.pfm")
    ui = UniformIntegrator(ladybug, UniformIntegratorSettings(frameSize=(192, 192), samplesPerPixel=64, maxWalkingDepth=48,
                                                              epsilonShell=1.0))
    ui.solve()
    assert abs(float(field.mean()) - float(ui.solution.mean())) < 0.02 * abs(float(ui.solution.mean()))
    assert np.array_equal(export_scene.read_pfm(exp / "dirichlet_sdf.pfm")[:, 0], ui.renderDirichletSDF())
