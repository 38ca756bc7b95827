#!/usr/bin/env python3
"""Write a shipped scene fixture (data/scenes/<name>.npz) as the files the JSON-driven entry
point reads: model.obj, boundary.obj, color.json (reference schema, core/problem.cu:63-96)
and a conf.json in the reference's layout (data/ladybug/u.json).

usage: export_scene.py <scene> <out_dir> [--frame N] [--spp N] [--depth N]
Floats are printed with 9 significant digits, which round-trips fp32 exactly."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elaina_amd import Problem  # noqa: E402


def write_obj(path, verts, segs):
    with open(path, "w") as f:
        for x, y in verts:
            f.write("v %.9g %.9g 0\n" % (x, y))
        for a, b in segs:
            f.write("l %d %d\n" % (a + 1, b + 1))


def write_colors(path, colors):
    cfg = [{"vertexID": i + 1,
            "leftColor": {"R": float(c[0]), "G": float(c[1]), "B": float(c[2])},
            "rightColor": {"R": float(c[3]), "G": float(c[4]), "B": float(c[5])}} for i, c in enumerate(colors)]
    with open(path, "w") as f:
        json.dump({"ColorConfigurations": cfg}, f)


# the "network" section of the reference's guided configurations (data/ladybug/n.json:49-81)
NETWORK_SECTION = {
    "encoding": {"otype": "DenseGrid", "interpolation": "Linear", "n_levels": 8, "n_features_per_level": 4,
                 "base_resolution": 8, "per_level_scale": 1.4049999713897705},
    "loss": {"otype": "L2"},
    "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64,
                "n_hidden_layers": 3},
    "optimizer": {"otype": "Ema", "decay": 0.949999988079071,
                  "nested": {"otype": "Adam", "adabound": False, "beta1": 0.8999999761581421, "beta2": 0.9900000095367432,
                             "epsilon": 1.0000000036274937e-15, "l2_reg": 9.999999974752427e-07,
                             "learning_rate": 0.00800000037997961}},
}


def export(scene, out_dir, frame=128, spp=16, depth=32, exp_name=None, integrator="uniform", train_spp=None, source=None):
    p = Problem.load_scene(scene)
    os.makedirs(out_dir, exist_ok=True)
    write_obj(os.path.join(out_dir, "model.obj"), p.d_verts, p.d_segs)
    write_obj(os.path.join(out_dir, "boundary.obj"), p.n_verts, p.n_segs)
    write_colors(os.path.join(out_dir, "color.json"), p.d_colors)
    conf = {
        "dimensionality": 2, "base_path": os.path.join(out_dir, "exp"), "exp_name": exp_name or (scene + "_u"),
        "integrator": {
            "setting": {"debugPixel": 0, "frameSize": [frame, frame], "maxWalkingDepth": depth, "samplesPerPixel": spp,
                        "saveSppMetricsDuration": -1, "saveSppMetricsUntil": -1, "saveTimeMetricsDuration": -1,
                        "epsilonShell": float(p.default_eps)},
            "type": "uniform", "channels": ["SOLUTION", "DIRICHLET_SDF"]},
        "export": [{"type": "image", "channel": "SOLUTION", "file_name": "solution"},
                   {"type": "image", "channel": "DIRICHLET_SDF", "file_name": "dirichlet_sdf"},
                   {"type": "energy", "tone": "NONE_NORMALIZED", "channel": "SOLUTION", "file_name": "solution_energy"}],
        "scene": {
            "aabb": {"min": [float(p.aabb[0]), float(p.aabb[1])], "max": [float(p.aabb[2]), float(p.aabb[3])]},
            "evaluation_grid": {"mData": {"pos": [float(p.probe[1]), float(p.probe[2])], "scale": float(p.probe[0]),
                                          "up": [float(p.probe[3]), float(p.probe[4])]}},
            "mesh": {"dirichlet_path": os.path.join(out_dir, "model.obj"),
                     "vertex_color_dirichlet_path": os.path.join(out_dir, "color.json"),
                     "neumann_path": os.path.join(out_dir, "boundary.obj")}}}
    if source is not None:
        # dense source grid (this build's stand-in for the reference's nanovdb "source_path")
        rgb = np.ascontiguousarray(source["rgb"], dtype="<f4")
        rgb.tofile(os.path.join(out_dir, "source.raw"))
        conf["scene"]["source_grid"] = {"path": os.path.join(out_dir, "source.raw"), "nx": int(rgb.shape[1]), "ny": int(rgb.shape[0]),
                                        "index_scale": [float(v) for v in source["index_scale"]],
                                        "index_offset": [float(v) for v in source["index_offset"]]}
        conf["scene"]["source_intensity"] = float(source.get("intensity", 1.0))
        conf["integrator"]["channels"].append("SOURCE")
        conf["export"].append({"type": "image", "channel": "SOURCE", "file_name": "source"})
    if integrator == "guided":
        conf["exp_name"] = exp_name or (scene + "_n")
        conf["integrator"]["type"] = "guided"
        conf["integrator"]["setting"].update({
            "trainSppCount": spp if train_spp is None else train_spp, "uniformFractionInTrainingPhase": 0.5,
            "uniformFractionInGuidingPhase": 0.5, "maxGuidedDepthInTrainingPhase": 10, "maxGuidedDepthInGuidingPhase": 10})
        conf["network"] = NETWORK_SECTION
        conf["print_network"] = True
    path = os.path.join(out_dir, "conf.json")
    with open(path, "w") as f:
        json.dump(conf, f, indent=4)
    return path


def export3(sd, out_dir, frame=(32, 32), spp=8, depth=64, eps=2e-3, exp_name="scene3d"):
    """a 3-D scene dict (tests/conftest.py cube_scene3 / sphere_scene3) as model.obj (v / f records), boundary.obj,
    colour files and a conf.json with "dimensionality": 3 (reference exec.cu:102-122, core/evaluation_grid.h:43-70)"""
    os.makedirs(out_dir, exist_ok=True)

    def write_obj3(path, verts, tris):
        with open(path, "w") as f:
            for x, y, z in verts:
                f.write("v %.9g %.9g %.9g\n" % (x, y, z))
            for a, b, c in tris:
                f.write("f %d %d %d\n" % (a + 1, b + 1, c + 1))
    mesh = {}
    if sd.get("d_tris") is not None:
        write_obj3(os.path.join(out_dir, "model.obj"), sd["d_verts"], sd["d_tris"])
        write_colors(os.path.join(out_dir, "color.json"), sd["d_colors"])
        mesh.update({"dirichlet_path": os.path.join(out_dir, "model.obj"), "vertex_color_dirichlet_path": os.path.join(out_dir, "color.json")})
    if sd.get("n_tris") is not None:
        write_obj3(os.path.join(out_dir, "boundary.obj"), sd["n_verts"], sd["n_tris"])
        write_colors(os.path.join(out_dir, "color_n.json"), sd["n_colors"])
        mesh.update({"neumann_path": os.path.join(out_dir, "boundary.obj"), "vertex_color_neumann_path": os.path.join(out_dir, "color_n.json")})
    scale, pos, up, right = sd["probe"]
    conf = {
        "dimensionality": 3, "base_path": os.path.join(out_dir, "exp"), "exp_name": exp_name,
        "integrator": {
            "setting": {"debugPixel": 0, "frameSize": [int(frame[0]), int(frame[1])], "maxWalkingDepth": depth, "samplesPerPixel": spp,
                        "saveSppMetricsDuration": -1, "saveSppMetricsUntil": -1, "saveTimeMetricsDuration": -1, "epsilonShell": float(eps)},
            "type": "uniform", "channels": ["SOLUTION", "DIRICHLET_SDF", "NEUMANN_SDF"]},
        "export": [{"type": "image", "channel": "SOLUTION", "file_name": "solution"},
                   {"type": "image", "channel": "DIRICHLET_SDF", "file_name": "dirichlet_sdf"},
                   {"type": "image", "channel": "NEUMANN_SDF", "file_name": "neumann_sdf"}],
        "scene": {"evaluation_grid": {"mData": {"pos": [float(v) for v in pos], "scale": float(scale), "up": [float(v) for v in up],
                                                "right": [float(v) for v in right]}},
                  "mesh": mesh}}
    src = sd.get("source")
    if src is not None:
        import numpy as np
        rgb = np.ascontiguousarray(src["rgb"], dtype="<f4")
        rgb.tofile(os.path.join(out_dir, "source.f32"))
        conf["scene"]["source_grid"] = {"path": os.path.join(out_dir, "source.f32"), "nx": int(rgb.shape[2]), "ny": int(rgb.shape[1]),
                                        "nz": int(rgb.shape[0]), "index_scale": [float(v) for v in src["index_scale"]],
                                        "index_offset": [float(v) for v in src["index_offset"]]}
        conf["scene"]["source_intensity"] = float(src.get("intensity", 1.0))
        conf["integrator"]["channels"].append("SOURCE")
        conf["export"].append({"type": "image", "channel": "SOURCE", "file_name": "source"})
    path = os.path.join(out_dir, "conf.json")
    with open(path, "w") as f:
        json.dump(conf, f, indent=4)
    return path


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = [int(x) for x in f.readline().split()]
        scale = float(f.readline())
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4")
    return data.reshape(h * w, 3)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("out_dir")
    ap.add_argument("--frame", type=int, default=128)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--depth", type=int, default=32)
    ap.add_argument("--integrator", default="uniform", choices=["uniform", "guided"])
    a = ap.parse_args()
    print(export(a.scene, a.out_dir, a.frame, a.spp, a.depth, integrator=a.integrator))
