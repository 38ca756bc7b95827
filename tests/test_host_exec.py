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
    for ext in (".exr", ".png", ".pfm"):
        assert os.path.exists(exp / ("solution" + ext)) and os.path.exists(exp / ("solution_energy" + ext))
    png = _read_png(exp / "solution.png")
    assert np.array_equal(png[::-1, :, :3].reshape(-1, 3), np.clip((field * np.float32(255)).astype(np.int32), 0, 255))


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


def _read_png(path):
    import struct
    import zlib
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, size = 8, b"", None
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        crc = struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0]
        assert zlib.crc32(typ + data) & 0xffffffff == crc
        if typ == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", data[:10])
            assert (depth, ctype) == (8, 6)
            size = (w, h)
        elif typ == b"IDAT":
            idat += data
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(size[1], 1 + 4 * size[0])
    assert np.all(raw[:, 0] == 0)
    return raw[:, 1:].reshape(size[1], size[0], 4)


def _read_exr_half_rgba(path):
    import struct
    b = open(path, "rb").read()
    assert struct.unpack("<ii", b[:8]) == (20000630, 2)
    pos, attrs = 8, {}
    while b[pos] != 0:
        e = b.index(b"\0", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\0", pos); typ = b[pos:e].decode(); pos = e + 1
        n = struct.unpack("<i", b[pos:pos + 4])[0]; pos += 4
        attrs[name] = (typ, b[pos:pos + n]); pos += n
    pos += 1
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0"
    names = [c[:1].decode() for c in attrs["channels"][1][:-1].split(b"\0")[::1] if len(c) == 1 and c.isalpha()]
    assert names[:4] == ["A", "B", "G", "R"]
    offs = struct.unpack("<%dQ" % h, b[pos:pos + 8 * h])
    img = np.zeros((h, w, 4), np.float32)
    for y in range(h):
        yy, nb = struct.unpack("<ii", b[offs[y]:offs[y] + 8])
        line = np.frombuffer(b[offs[y] + 8:offs[y] + 8 + nb], dtype="<f2").reshape(4, w)
        img[yy] = line[[3, 2, 1, 0]].T.astype(np.float32)          # -> R G B A
    return img


def test_png_and_exr_writers_and_colormaps(tmp_path):
    """reference core/texture.cu:82-116: PNG = clamp((int)(v*255)), RGBA8, flipped vertically;
    EXR = half RGBA, flipped; colormaps of util/tonemapping.cuh (JET formula, PARULA / RDBU fitted tables)"""
    out = subprocess.run([_exe(), "--imagetest", str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    w, h = 5, 3
    ys, xs = np.mgrid[0:h, 0:w]
    want = np.stack([xs / 4.0, ys - 0.5, 0.1 * (xs + ys * w)], -1).astype(np.float32)
    png = _read_png(tmp_path / "grad.png")
    q = np.clip((want[::-1] * np.float32(255)).astype(np.int32), 0, 255)
    assert np.array_equal(png[..., :3], q) and np.all(png[..., 3] == 255)
    exr = _read_exr_half_rgba(tmp_path / "grad.exr")
    assert np.array_equal(exr[..., :3], want[::-1].astype(np.float16).astype(np.float32)) and np.all(exr[..., 3] == 1)
    tones = {}
    for line in out.stdout.splitlines():
        f = line.split()
        if f[0] == "tone":
            tones[(int(f[1]), float(f[2]))] = np.array([float(v) for v in f[3:]])
    # MATLAB_JET = 2: dark blue -> cyan/green -> yellow -> dark red
    np.testing.assert_allclose(tones[(2, 0.0)], [0, 0, 0.5], atol=1e-6)
    np.testing.assert_allclose(tones[(2, 0.5)], [0.5, 1.0, 0.5], atol=1e-6)
    np.testing.assert_allclose(tones[(2, 1.0)], [0.5, 0, 0], atol=1e-6)
    # IDL_RDBU = 4: the reference's fitted table (util/tonemapping.cuh:385-480) runs through the ColorBrewer
    # RdBu-11 colours: #67001f at 0, #f7f7f7 in the middle, #053061 at 1
    np.testing.assert_allclose(tones[(4, 0.0)], np.array([103, 0, 31]) / 255.0, atol=0.01)
    np.testing.assert_allclose(tones[(4, 0.5)], np.array([247, 247, 247]) / 255.0, atol=0.01)
    np.testing.assert_allclose(tones[(4, 1.0)], np.array([5, 48, 97]) / 255.0, atol=0.015)
    # MATLAB_PARULA = 3 (util/tonemapping.cuh:53-383): cubic pieces through MATLAB's parula(64) table, whose first
    # and last rows are (0.2081, 0.1663, 0.5292) and (0.9763, 0.9831, 0.0538)
    # (the first knot sits at x = 1/64: x = 0 is the first cubic extrapolated half a step), the middle one (0.1801, 0.7177, 0.6424)
    np.testing.assert_allclose(tones[(3, 0.0)], [0.2081, 0.1663, 0.5292], atol=0.06)
    np.testing.assert_allclose(tones[(3, 0.5)], [0.1801, 0.7177, 0.6424], atol=2e-3)
    np.testing.assert_allclose(tones[(3, 1.0)], [0.9763, 0.9831, 0.0538], atol=1e-5)
    np.testing.assert_allclose(tones[(1, 0.25)], [0.25] * 3)


@pytest.mark.gpu
def test_run_expr_with_a_source_grid(tmp_path, oracle, ladybug):
    """scene.source_grid (dense stand-in for the nanovdb source_path) through the C++ host:
    Poisson solve and the SOURCE channel, both equal to the oracle"""
    import copy
    import export_scene
    n = 33
    gx, gy = np.meshgrid(np.linspace(0, 1, n), np.linspace(0, 1, n))
    src = {"rgb": np.stack([np.sin(4 * gx) * gy, gx, 1 - gy], -1).astype(np.float32), "index_scale": (32 / 700.0, 32 / 700.0),
           "index_offset": (100 * 32 / 700.0, 100 * 32 / 700.0), "intensity": 2e-3}
    conf = export_scene.export("ladybug", str(tmp_path), frame=48, spp=4, depth=32, source=src)
    out = subprocess.run([_exe(), conf], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    exp = tmp_path / "exp" / "ladybug_u"
    p = copy.copy(ladybug)
    p.source = {"rgb": src["rgb"], "index_scale": src["index_scale"], "index_offset": src["index_offset"], "intensity": 2e-3}
    ref = oracle.solve(p.as_dict(), 48, 48, 4, 32, 1.0, threads=os.cpu_count())
    assert np.array_equal(export_scene.read_pfm(exp / "solution.pfm"), ref["field"])
    assert np.array_equal(export_scene.read_pfm(exp / "source.pfm"), oracle.render_source(p.as_dict(), 48, 48))
    plain = oracle.solve(ladybug.as_dict(), 48, 48, 4, 32, 1.0, threads=os.cpu_count())
    assert not np.array_equal(plain["field"], ref["field"])


def _write_png(path, img, ctype, level=9, strategy=0, palette=None, filters=None):
    """minimal PNG writer for the reader's tests: img uint8 [h, w, channels]; per-row filter types"""
    import struct
    import zlib
    h, w, ch = img.shape
    rows = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        ft = (filters[y % len(filters)] if filters else 0)
        left = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        ul = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        if ft == 0:
            pred = np.zeros_like(cur)
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) >> 1
        else:
            p = left + prev - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
        rows.append(ft)
        rows += bytes(((cur - pred) & 255).astype(np.uint8))
        prev = cur
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
    z = co.compress(bytes(rows)) + co.flush()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette.reshape(-1)))
    # split IDAT into two chunks on purpose
    out += chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")
    open(path, "wb").write(out)


def test_png_reader_and_mask_loading(tmp_path):
    """read_png (mask images, reference core/problem.cu:216-242 via stb_image): stored / fixed /
    dynamic deflate blocks, all five filters, grey / RGB / RGBA / grey+alpha / palette"""
    import zlib
    rng = np.random.default_rng(1)
    exe = _exe()
    cases = [("rgb", 2, 3, 9, 0), ("rgba", 6, 4, 6, 0), ("grey", 0, 1, 0, 0), ("ga", 4, 2, 9, zlib.Z_FIXED), ("pal", 3, 1, 9, 0)]
    for name, ctype, ch, level, strategy in cases:
        w, h = 37, 23
        # smooth + noisy content so that every filter and long back-references occur
        base = (np.add.outer(np.arange(h) * 3, np.arange(w) * 5)[..., None] + np.arange(ch) * 40) % 256
        img = np.where(rng.uniform(size=(h, w, ch)) < 0.2, rng.integers(0, 256, (h, w, ch)), base).astype(np.uint8)
        palette = rng.integers(0, 256, (256, 3)).astype(np.uint8) if ctype == 3 else None
        _write_png(tmp_path / (name + ".png"), img, ctype, level, strategy, palette, filters=[0, 1, 2, 3, 4])
        out = subprocess.run([exe, "--readpng", str(tmp_path / (name + ".png")), str(tmp_path / (name + ".raw"))],
                             capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.split() == [str(w), str(h)], out.stderr
        got = np.fromfile(tmp_path / (name + ".raw"), dtype=np.uint8).reshape(h, w, 4)
        if ctype == 2:
            want = np.concatenate([img, np.full((h, w, 1), 255, np.uint8)], -1)
        elif ctype == 6:
            want = img
        elif ctype == 0:
            want = np.concatenate([img.repeat(3, -1), np.full((h, w, 1), 255, np.uint8)], -1)
        elif ctype == 4:
            want = np.concatenate([img[..., :1].repeat(3, -1), img[..., 1:]], -1)
        else:
            want = np.concatenate([palette[img[..., 0]], np.full((h, w, 1), 255, np.uint8)], -1)
        assert np.array_equal(got, want), name
    bad = subprocess.run([exe, "--readpng", str(tmp_path / "missing.png"), str(tmp_path / "x.raw")], capture_output=True, text=True)
    assert bad.returncode == 1 and "cannot open" in bad.stderr
    # malformed headers are refused before any size is computed from them: a short IHDR, an IHDR that
    # is not the first chunk, absurd dimensions, a file cut inside a chunk
    import struct
    good = open(tmp_path / "rgb.png", "rb").read()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    ihdr = good[16:29]
    rest = good[33:]
    for name, blob, msg in [
        ("short_ihdr", good[:8] + chunk(b"IHDR", ihdr[:9]) + rest, "IHDR"),
        ("late_ihdr", good[:8] + chunk(b"tEXt", b"a\0b") + chunk(b"IHDR", ihdr) + rest, "IHDR"),
        ("huge", good[:8] + chunk(b"IHDR", struct.pack(">II", 1 << 30, 1 << 30) + ihdr[8:]) + rest, "dimensions"),
        ("cut", good[:40], "png:"),
    ]:
        open(tmp_path / (name + ".png"), "wb").write(blob)
        out = subprocess.run([exe, "--readpng", str(tmp_path / (name + ".png")), str(tmp_path / "x.raw")], capture_output=True, text=True)
        assert out.returncode == 1 and msg in out.stderr, (name, out.stderr)


def _write_exr(path, img, compression, half):
    """single-part scan-line OpenEXR with channels B G R (alphabetical), compression 0 NONE / 2 ZIPS / 3 ZIP"""
    import struct, zlib
    h, w, _ = img.shape
    hdr = struct.pack("<ii", 20000630, 2)

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chl = b"".join(c + b"\0" + struct.pack("<iBBBBii", 1 if half else 2, 0, 0, 0, 0, 1, 1) for c in (b"B", b"G", b"R")) + b"\0"
    hdr += attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression]))
    hdr += attr("dataWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1)) + attr("displayWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1))
    hdr += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    per = 16 if compression == 3 else 1
    blocks = []
    for y0 in range(0, h, per):
        raw = b""
        for y in range(y0, min(y0 + per, h)):
            for c in (2, 1, 0):            # B, G, R
                raw += img[y, :, c].astype(np.float16 if half else np.float32).tobytes()
        if compression:
            t = np.frombuffer(raw, np.uint8)
            t = np.concatenate([t[0::2], t[1::2]]).astype(np.int32)
            d = t.copy()
            d[1:] = (t[1:] - t[:-1] + 128 + 256) & 255
            z = zlib.compress(bytes(d.astype(np.uint8)))
            raw = z if len(z) < len(raw) else raw
        blocks.append((y0, raw))
    table_at = len(hdr)
    pos = table_at + 8 * len(blocks)
    offs, body = b"", b""
    for y0, raw in blocks:
        offs += struct.pack("<Q", pos)
        body += struct.pack("<ii", y0, len(raw)) + raw
        pos += 8 + len(raw)
    open(path, "wb").write(hdr + offs + body)


def test_mask_images_of_every_readable_format(tmp_path):
    """mask_path goes through Image::loadImage in the reference (core/problem.cu:216-242, core/texture.cu:26-80): PNG, OpenEXR
    (uncompressed, ZIPS, ZIP; half and float), PFM (colour and grey, both byte orders) and Radiance .hdr (flat pixels and
    run-length scan lines) give the same mask here; JPEG -- lossy: the mask would hang on the last bit of stb_image's own inverse
    transform -- is out of scope and refused with a message that says so"""
    rng = np.random.default_rng(3)
    exe = _exe()
    w, h = 41, 35
    img = np.zeros((h, w, 3), np.float32)
    on = rng.uniform(size=(h, w)) < 0.6
    img[on] = rng.uniform(0.01, 2.0, size=(int(on.sum()), 3)).astype(np.float32)
    img[3, 5] = (0.0, 0.0, 0.25)          # a single non-zero channel is enough
    on[3, 5] = True
    want = on[::-1].astype(np.uint8)      # loaded flipped vertically

    def mask_of(name):
        out = subprocess.run([exe, "--readmask", str(tmp_path / name), str(tmp_path / "m.raw")], capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.split() == [str(w), str(h)], (name, out.stderr)
        return np.fromfile(tmp_path / "m.raw", dtype=np.uint8).reshape(h, w)

    _write_png(tmp_path / "m.png", np.ceil(np.clip(img, 0, 1) * 255).astype(np.uint8), 2, filters=[1, 4])
    assert np.array_equal(mask_of("m.png"), want)
    for comp in (0, 2, 3):
        for half in (True, False):
            name = "m_%d_%d.exr" % (comp, half)
            _write_exr(tmp_path / name, img, comp, half)
            assert np.array_equal(mask_of(name), want), name
    for big_endian in (False, True):
        # PFM rows run bottom to top: the file holds the image flipped, the load flips it back
        data = img[::-1].astype(">f4" if big_endian else "<f4").tobytes()
        open(tmp_path / "m.pfm", "wb").write(b"PF\n%d %d\n%s\n" % (w, h, b"1.0" if big_endian else b"-1.0") + data)
        assert np.array_equal(mask_of("m.pfm"), want)
    grey = img[::-1].max(-1).astype("<f4").tobytes()
    open(tmp_path / "g.pfm", "wb").write(b"Pf\n%d %d\n-1.0\n" % (w, h) + grey)
    assert np.array_equal(mask_of("g.pfm"), want)
    # the exporter's own EXR files read back (half RGBA, uncompressed): grad.exr of --imagetest has a zero at (0, 0) only where all channels vanish
    out = subprocess.run([exe, "--imagetest", str(tmp_path / "img")], capture_output=True, text=True)
    assert out.returncode == 0
    got = subprocess.run([exe, "--readmask", str(tmp_path / "img" / "grad.exr"), str(tmp_path / "m.raw")], capture_output=True, text=True)
    assert got.returncode == 0 and got.stdout.split() == ["5", "3"], got.stderr
    # Radiance RGBE: mantissa bytes and a shared exponent; a pixel is black exactly when its exponent byte is zero
    mant, ex = np.frexp(img.max(-1))
    e8 = np.where(img.max(-1) > 1e-32, ex + 128, 0).astype(np.int32)
    scale = np.where(e8 > 0, np.ldexp(1.0, 8 - ex.astype(np.int32)), 0.0)
    rgbe = np.concatenate([np.minimum(np.floor(img * scale[..., None]), 255), e8[..., None]], -1).astype(np.uint8)
    rgbe[on & (rgbe[..., :3].max(-1) == 0)] = (0, 0, 1, 100)     # (a mantissa that rounded to zero would switch the pixel off)
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y %d +X %d\n" % (h, w)
    open(tmp_path / "flat.hdr", "wb").write(head + rgbe.tobytes())
    assert np.array_equal(mask_of("flat.hdr"), want)

    def rle(row):      # one channel of a scan line: runs of >= 3 equal bytes as runs, the rest as literals
        out, i = bytearray(), 0
        while i < len(row):
            j = i
            while j < len(row) and j - i < 127 and row[j] == row[i]:
                j += 1
            if j - i >= 3:
                out += bytes([128 + j - i, row[i]])
                i = j
            else:
                k = i
                while k < len(row) and k - i < 127 and not (k + 2 < len(row) and row[k] == row[k + 1] == row[k + 2]):
                    k += 1
                k = max(k, i + 1)
                out += bytes([k - i]) + bytes(row[i:k])
                i = k
        return bytes(out)

    body = b"".join(bytes([2, 2, w >> 8, w & 255]) + b"".join(rle(rgbe[y, :, c].tolist()) for c in range(4)) for y in range(h))
    open(tmp_path / "rle.hdr", "wb").write(head.replace(b"#?RADIANCE", b"#?RGBE") + body)
    assert np.array_equal(mask_of("rle.hdr"), want)
    open(tmp_path / "cut.hdr", "wb").write((head + body)[:200])
    bad = subprocess.run([exe, "--readmask", str(tmp_path / "cut.hdr"), str(tmp_path / "m.raw")], capture_output=True, text=True)
    assert bad.returncode == 1 and "hdr" in bad.stderr
    # a hundred bytes that claim 2^24 x 2^24 pixels: the clean "truncated file", not an allocation of 2^50 bytes
    open(tmp_path / "huge.hdr", "wb").write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 16777216 +X 16777216\n" + b"\2\2\0\0" * 8)
    bad = subprocess.run([exe, "--readmask", str(tmp_path / "huge.hdr"), str(tmp_path / "m.raw")], capture_output=True, text=True)
    assert bad.returncode == 1 and "truncated file" in bad.stderr
    open(tmp_path / "x.jpg", "wb").write(b"\xff\xd8\xff\xe0" + b"\0" * 64)
    bad = subprocess.run([exe, "--readmask", str(tmp_path / "x.jpg"), str(tmp_path / "m.raw")], capture_output=True, text=True)
    assert bad.returncode == 1 and "out of scope" in bad.stderr and "PNG" in bad.stderr
    open(tmp_path / "cut.exr", "wb").write(open(tmp_path / "m_3_1.exr", "rb").read()[:300])
    bad = subprocess.run([exe, "--readmask", str(tmp_path / "cut.exr"), str(tmp_path / "m.raw")], capture_output=True, text=True)
    assert bad.returncode == 1 and "exr" in bad.stderr


@pytest.mark.gpu
def test_run_expr_with_a_mask_image(tmp_path, oracle, ladybug):
    """scene.mask_path through the C++ host: flipped vertically, on = any non-zero RGB byte"""
    import copy
    import export_scene
    conf = export_scene.export("ladybug", str(tmp_path), frame=40, spp=3, depth=24)
    rng = np.random.default_rng(2)
    img = (rng.uniform(size=(40, 40, 3)) < 0.25).astype(np.uint8) * rng.integers(1, 256, (40, 40, 3)).astype(np.uint8)
    _write_png(tmp_path / "mask.png", img, 2, filters=[4, 1])
    c = json.load(open(conf))
    c["scene"]["mask_path"] = str(tmp_path / "mask.png")
    json.dump(c, open(conf, "w"))
    out = subprocess.run([_exe(), conf], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    p = copy.copy(ladybug)
    p.mask = (img[::-1].max(-1) != 0).astype(np.uint8).reshape(-1)
    ref = oracle.solve(p.as_dict(), 40, 40, 3, 24, 1.0)
    field = export_scene.read_pfm(tmp_path / "exp" / "ladybug_u" / "solution.pfm")
    assert np.array_equal(field, ref["field"]) and np.all(field[p.mask == 0] == 0) and 0 < p.mask.sum() < 1600


@pytest.mark.gpu
def test_run_expr_spp_metric_frames(tmp_path, oracle, ladybug):
    """saveSppMetricsDuration / Until (reference integrator.cu:578-592, guided :1049-1063): frame k
    holds the solution after k + 1 samples -- for the uniform integrator that is the oracle's solve
    with spp = k + 1, bit for bit (8-bit PNG quantisation applied)"""
    import export_scene
    for integrator in ("uniform", "guided"):
        d = tmp_path / integrator
        conf = export_scene.export("ladybug", str(d), frame=32, spp=5, depth=24, integrator=integrator, train_spp=0)
        c = json.load(open(conf))
        c["integrator"]["setting"].update({"saveSppMetricsDuration": 2, "saveSppMetricsUntil": 4})
        c["integrator"]["setting"]["saveTimeMetricsDuration"] = 4      # both integrators (uniform: integrator.cu:594-609)
        json.dump(c, open(conf, "w"))
        out = subprocess.run([_exe(), conf], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        exp = d / "exp" / ("ladybug_u" if integrator == "uniform" else "ladybug_n")
        frames = sorted(os.listdir(exp / "frames"))
        assert frames == ["0.exr", "0.png", "2.exr", "2.png"]          # sampleId 0 and 2 (< until 4), not 4
        if integrator == "uniform":
            for k in (0, 2):
                ref = oracle.solve(ladybug.as_dict(), 32, 32, k + 1, 24, 1.0)["field"]
                png = _read_png(exp / "frames" / ("%d.png" % k))
                assert np.array_equal(png[::-1, :, :3].reshape(-1, 3), np.clip((ref * np.float32(255)).astype(np.int32), 0, 255))
            final = oracle.solve(ladybug.as_dict(), 32, 32, 5, 24, 1.0)["field"]
            assert np.array_equal(export_scene.read_pfm(exp / "solution.pfm"), final)      # spp restored afterwards
            # frames_time/<elapsed ms>.png: samples 0 and 4; the later one is the solution after 5 samples
            names = sorted(os.listdir(exp / "frames_time"), key=lambda n_: int(n_.split(".")[0]))
            assert len(names) in (2, 4) and all(n_.split(".")[0].isdigit() for n_ in names)
            last = _read_png(exp / "frames_time" / [n_ for n_ in names if n_.endswith(".png")][-1])
            assert np.array_equal(last[::-1, :, :3].reshape(-1, 3), np.clip((final * np.float32(255)).astype(np.int32), 0, 255))
        else:
            assert len(os.listdir(exp / "frames_time")) in (2, 4)      # samples 0 and 4 (.exr + .png; names are elapsed ms)


@pytest.mark.gpu
def test_run_expr_three_dimensional_configuration(tmp_path, oracle):
    """"dimensionality": 3 through the C++ host (reference exec.cu:102-122): OBJ triangles + colour files ->
    Problem<3> -> UniformIntegrator<3> -> raw field, against the oracle; an unknown integrator type is refused"""
    import export_scene
    from conftest import cube_scene3
    sd = cube_scene3(n=2, d_faces=(0, 1), n_faces=(2, 3, 4, 5), value=lambda x, y, z: x + z, flux=lambda x, y, z, f: 0.0)
    conf = export_scene.export3(sd, str(tmp_path), frame=(24, 16), spp=6, depth=48, eps=2e-3)
    out = subprocess.run([_exe(), conf], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    exp = tmp_path / "exp" / "scene3d"
    res = json.load(open(exp / "result.json"))
    ref = oracle.solve3(sd, 24, 16, 6, 48, 2e-3, threads=os.cpu_count())
    assert res["walk_steps"] == ref["walk_steps"] and "duration" in res
    assert np.array_equal(export_scene.read_pfm(exp / "solution.pfm"), ref["field"])
    # the SDF channels of the 3-D integrator (renderDirichletSDF / renderSilhouetteSDF, integrator/common.h:52-123)
    assert np.array_equal(export_scene.read_pfm(exp / "dirichlet_sdf.pfm")[:, 0], oracle.render_sdf3(sd, 24, 16, 0))
    assert np.array_equal(export_scene.read_pfm(exp / "neumann_sdf.pfm")[:, 0], oracle.render_sdf3(sd, 24, 16, 1))
    cj = json.load(open(conf))
    cj["integrator"]["type"] = "nonesuch"
    json.dump(cj, open(conf, "w"))
    out = subprocess.run([_exe(), conf], capture_output=True, text=True)
    assert out.returncode == 1 and "integrator type" in out.stderr
    # a Poisson problem: the source term as a dense 3-D grid ("source_grid" with nz)
    from test_oracle_3d import _unit_source
    sd["source"] = dict(_unit_source(), intensity=0.75)
    conf = export_scene.export3(sd, str(tmp_path / "poisson"), frame=(16, 16), spp=5, depth=48, eps=2e-3)
    out = subprocess.run([_exe(), conf], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    ref = oracle.solve3(sd, 16, 16, 5, 48, 2e-3, threads=os.cpu_count())
    got = export_scene.read_pfm(tmp_path / "poisson" / "exp" / "scene3d" / "solution.pfm")
    assert np.array_equal(got, ref["field"])
    plain = oracle.solve3({k: v for k, v in sd.items() if k != "source"}, 16, 16, 5, 48, 2e-3, threads=4)["field"]
    assert np.mean(got[:, 0] - plain[:, 0]) > 0.01          # f > 0 raises the solution
    assert np.array_equal(export_scene.read_pfm(tmp_path / "poisson" / "exp" / "scene3d" / "source.pfm"), oracle.render_source3(sd, 16, 16))


@pytest.mark.gpu
def test_run_expr_three_dimensional_guided_configuration(tmp_path, oracle):
    """"dimensionality": 3 with "type": "guided" through the C++ host (reference exec.cu:102-122 dispatches GuidedIntegrator<3>;
    :175-186 print_network asks for the mixture at (0, -0.21, 0)): the network section, scene.aabb with three entries, the
    reference's training constants (one Adam step needs 65 536 records) -- field, counters and optimizer steps against the oracle"""
    import export_scene
    from conftest import cube_scene3
    from oracle.oracle import default_net_config3, guided_settings3
    sd = cube_scene3(n=2, d_faces=(4, 5), n_faces=(0, 1, 2, 3), value=lambda x, y, z: z, flux=lambda x, y, z, f: 0.0)
    w, h, spp, train, depth, eps = 176, 160, 3, 2, 32, 2e-3
    conf = export_scene.export3(sd, str(tmp_path), frame=(w, h), spp=spp, depth=depth, eps=eps)
    cj = json.load(open(conf))
    cj["integrator"]["type"] = "guided"
    cj["integrator"]["setting"].update({"trainSppCount": train, "uniformFractionInTrainingPhase": 0.5, "uniformFractionInGuidingPhase": 0.5,
                                        "maxGuidedDepthInTrainingPhase": 10, "maxGuidedDepthInGuidingPhase": 10})
    cj["scene"]["aabb"] = {"min": [-0.1, -0.1, -0.1], "max": [1.1, 1.1, 1.1]}
    net = dict(export_scene.NETWORK_SECTION)
    net["encoding"] = dict(net["encoding"], n_levels=4)          # a small dense grid (four levels), the same code path
    cj["network"] = net
    cj["print_network"] = True
    json.dump(cj, open(conf, "w"))
    out = subprocess.run([_exe(), conf], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "VMM @ (0.000000, -0.210000, 0.000000)" in out.stdout + out.stderr
    exp = tmp_path / "exp" / "scene3d"
    res = json.load(open(exp / "result.json"))
    # the oracle from the network the library initialises (seed 42): read it through the Python mirror
    from elaina_amd.guided import GuidedIntegratorSettings
    from elaina_amd.integrator3d import GuidedIntegrator3, Problem3, default_net_config3 as hip_cfg3
    st = GuidedIntegratorSettings(frameSize=(w, h), samplesPerPixel=spp, trainSppCount=train, maxWalkingDepth=depth, epsilonShell=eps)
    gi = GuidedIntegrator3(Problem3.from_dict(sd), st, ((-0.1, -0.1, -0.1), (1.1, 1.1, 1.1)), network_config=hip_cfg3(n_levels=4), seed=42)
    p0 = gi.network.params()
    gi.close()
    gs = guided_settings3(w, h, spp, depth, eps, (-0.1, -0.1, -0.1), (1.1, 1.1, 1.1), train_spp_count=train)
    ref = oracle.solve_guided3(sd, gs, default_net_config3(n_levels=4), p0.copy(), threads=os.cpu_count())
    assert res["walk_steps"] == ref["walk_steps"] and res["guided_steps"] == ref["guided_steps"] > 0
    assert res["optimizer_steps"] == ref["optimizer_steps"] >= 1
    assert np.array_equal(export_scene.read_pfm(exp / "solution.pfm"), ref["field"])
    # without scene.aabb the guided integrator cannot be built
    del cj["scene"]["aabb"]
    json.dump(cj, open(conf, "w"))
    out = subprocess.run([_exe(), conf], capture_output=True, text=True)
    assert out.returncode == 1 and "aabb" in out.stderr
