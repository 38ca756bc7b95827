"""Python mirror of the scene side of the boundary (reference core/problem.h:54-194).

Only what the hot path needs crosses the C-ABI: geometry, per-vertex colour pairs,
intensities, probe and mask.  The C++ host (elaina_amd/host/) is the primary mirror of
Problem<2>; this class exists so that tests and bench.py can drive the same C-ABI.
"""
import os

import numpy as np

_SCENES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data", "scenes")


class Problem:
    """2-D problem description (reference Problem<2>)."""

    def __init__(self, d_verts=None, d_segs=None, d_colors=None, n_verts=None, n_segs=None, n_colors=None,
                 probe=(1.0, 0.0, 0.0, 0.0, 1.0), dirichlet_intensity=1.0, neumann_intensity=1.0, mask=None,
                 aabb=None, source=None):
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
        i32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        self.d_verts, self.d_segs, self.d_colors = f32(d_verts), i32(d_segs), f32(d_colors)
        self.n_verts, self.n_segs, self.n_colors = f32(n_verts), i32(n_segs), f32(n_colors)
        self.probe = np.asarray(probe, dtype=np.float32)
        self.dirichlet_intensity = float(dirichlet_intensity)
        self.neumann_intensity = float(neumann_intensity)
        self.mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.aabb = aabb
        # source term f of laplace(u) = -f: dict(rgb=[ny, nx, 3], index_scale=(sx, sy), index_offset=(ox, oy),
        # intensity=1.0), index = world * scale + offset, bilinear, zero outside (stands in for the
        # reference's nanovdb grid, core/problem.cu:136-149)
        self.source = None
        if source is not None:
            self.source = {"rgb": np.ascontiguousarray(source["rgb"], dtype=np.float32),
                           "index_scale": tuple(float(v) for v in source["index_scale"]),
                           "index_offset": tuple(float(v) for v in source["index_offset"]),
                           "intensity": float(source.get("intensity", 1.0))}
            if self.source["rgb"].ndim != 3 or self.source["rgb"].shape[2] != 3:
                raise ValueError("source rgb must be [ny, nx, 3]")

    # reference getters (core/problem.h:104-111)
    def isDirichletEnabled(self):
        return self.d_segs is not None and len(self.d_segs) > 0

    def isNeumannEnabled(self):
        return self.n_segs is not None and len(self.n_segs) > 0

    def isSourceEnabled(self):
        return self.source is not None

    def as_dict(self):
        """plain dict of arrays (the oracle binding in tests takes the same dict)"""
        return {
            "d_verts": self.d_verts, "d_segs": self.d_segs, "d_colors": self.d_colors,
            "n_verts": self.n_verts, "n_segs": self.n_segs, "n_colors": self.n_colors,
            "probe": self.probe, "dirichlet_intensity": self.dirichlet_intensity,
            "neumann_intensity": self.neumann_intensity, "mask": self.mask, "source": self.source,
        }

    @staticmethod
    def scene_path(name):
        return os.path.join(_SCENES, name + ".npz")

    @classmethod
    def load_scene(cls, name):
        """Shipped fixtures: 'ladybug' or 'fille' (data/scenes/*.npz, tools/import_scenes.py)."""
        d = np.load(cls.scene_path(name))
        p = cls(d_verts=d["d_verts"], d_segs=d["d_segs"], d_colors=d["d_colors"], n_verts=d["n_verts"],
                n_segs=d["n_segs"], probe=d["probe"], aabb=d["aabb"])
        p.default_max_depth = int(d["settings"][0])
        p.default_eps = float(d["eps"][0])
        return p
