"""CPU tests of the glTF / GLB scene loader (rt::loadScene of the host library), the next row after the
hot path: a scene written by tests/gltf_writer.py must come back field for field, in every container;
the PNG and Radiance readers are checked against encoders / decoders restated here in numpy, and against
the image files the reference ships when they are present on this machine.

Parity status: the reference loads through fastgltf + stb_image, neither of which is available here, and
holds no loader tests or golden scenes; the field mapping is pinned by the citations in host/scene.cpp,
the decoders by the format definitions restated below ("parity unpinned" against upstream's binaries)."""
import json
import os
import struct
import zlib

import numpy as np
import pytest

from raytracer_amd import engine, workloads
from gltf_writer import encode_hdr, encode_png, float_to_rgbe, rgbe_to_float, write_scene


@pytest.fixture(scope="module")
def R():
    from raytracer_amd import renderer
    renderer.load()
    return renderer


def expect_same_scene(loaded, scene):
    counts = loaded.counts()
    assert counts["geometries"] == len(scene["geometries"])
    assert counts["instances"] == len(scene["instances"])
    assert counts["materials"] == len(scene["materials"])
    assert counts["cameras"] == len(scene["cameras"]) and counts["lights"] == len(scene["lights"])
    for k, g in enumerate(scene["geometries"]):
        got = loaded.geometry(k)
        v = np.asarray(g["vertices"], dtype=np.float32)
        packed = np.concatenate([v[:, 0:3], v[:, 3:6], v[:, 8:12], v[:, 12:14], v[:, 14:16]], axis=1)
        assert np.array_equal(got["vertices"].view(np.uint32), packed.view(np.uint32))
        assert np.array_equal(got["indices"], np.asarray(g["indices"], dtype=np.uint32).reshape(-1, 3))
        assert got["material"] == g["material"] and got["name"] == f"mesh.0"
    for k, (gi, m) in enumerate(scene["instances"]):
        geometry, matrix = loaded.instance(k)
        assert geometry == gi and np.array_equal(matrix, np.asarray(m, dtype=np.float32).reshape(-1))
    from raytracer_amd import renderer as Rm
    for k, m in enumerate(scene["materials"]):
        got, cutoff, flags = loaded.material(k)
        want = Rm.gpu_material(m)
        assert bytes(got)[:264] == bytes(want)[:264], k
        assert cutoff == 0.5 and flags == 0
    for k, c in enumerate(scene["cameras"]):
        got = loaded.camera(k)
        assert np.array_equal(got["transform"], np.asarray(c["transform"], np.float32).reshape(-1))
        for key in ("verticalFov", "aspectRatio", "nearPlane", "aperture", "focusDistance"):
            assert np.float32(got[key]) == np.float32(c[key]), key
        assert got["farPlane"] == float("inf")
    for k, l in enumerate(scene["lights"]):
        got = loaded.light(k)
        assert np.array_equal(got["transform"], np.asarray(l["transform"], np.float32).reshape(-1))
        assert np.array_equal(got["color"], np.asarray(l["color"], np.float32)) and got["type"] == l["type"]
        assert np.float32(got["intensity"]) == np.float32(l["intensity"])
        assert np.float32(got["innerCone"]) == np.float32(l["innerCone"]) if l["type"] == 1 else True


@pytest.mark.parametrize("container", ["gltf", "embedded", "glb"])
def test_cornell_round_trip(R, tmp_path, container):
    for light in ("point", "panel"):
        scene = workloads.cornell_box(light=light)
        path = write_scene(scene, tmp_path / f"cornell_{light}.{'glb' if container == 'glb' else 'gltf'}", container)
        expect_same_scene(R.load_scene(path), scene)


def test_textures_environment_and_material_extensions(R, tmp_path):
    scene = workloads.textured_sky_scene()
    rgbe = float_to_rgbe(np.asarray(scene["environment"]["pixels"], dtype=np.float32)[..., :3])
    (tmp_path / "sky.hdr").write_bytes(encode_hdr(rgbe))
    path = write_scene(scene, tmp_path / "sky.glb", "glb", environment_file="sky.hdr")
    loaded = R.load_scene(path)
    expect_same_scene(loaded, scene)
    counts = loaded.counts()
    assert counts["images"] == 1 and counts["textures"] == 1
    image = loaded.image(0)
    assert image["mimeType"] == 2 and image["encoded_bytes"] > 0
    assert np.array_equal(image["rgba"], np.asarray(scene["images"][0], dtype=np.uint8))
    assert loaded.texture(0) == dict(image=0, magFilter=0, minFilter=0, wrapU=10497, wrapV=10497)
    env = loaded.environment()
    assert (env["width"], env["height"]) == (rgbe.shape[1], rgbe.shape[0])
    assert np.float32(env["rotation"]) == np.float32(0.4) and np.float32(env["strength"]) == np.float32(1.2)
    assert np.float32(env["exposure"]) == np.float32(-0.5)
    assert np.array_equal(env["pixels"], rgbe_to_float(rgbe))


def test_node_hierarchy_trs_quantized_and_generated_indices(R, tmp_path):
    """Children inherit their parents' transforms (T * R * S, depth first, in node order), normalised
    integer attributes decode by the KHR_mesh_quantization rules, and a primitive without indices gets
    0, 1, 2, ...; the default material is index 0 and glTF material i becomes i + 1."""
    positions = np.array([[0, 0, 0], [127, 0, 0], [0, 127, 0], [0, 0, 127], [127, 127, 0], [0, 127, 127]], dtype=np.int8)
    normals = np.array([[0, 0, 32767], [0, 0, -32767], [32767, 0, 0], [0, 0, -32768], [0, 32767, 0], [0, 0, 32767]],
                       dtype=np.int16)
    uvs = np.array([[0, 0], [65535, 0], [0, 65535], [255, 255], [1, 2], [3, 4]], dtype=np.uint16)
    blob = bytearray()
    blob += positions.tobytes() + b"\0" * 2      # 18 bytes + pad -> 20
    blob += normals.tobytes()                     # 36 bytes at 20
    blob += uvs.tobytes()                         # 24 bytes at 56
    doc = {
        "asset": {"version": "2.0"}, "extensionsUsed": ["KHR_mesh_quantization"],
        "extensionsRequired": ["KHR_mesh_quantization"],
        "buffers": [{"byteLength": len(blob)}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 18, "byteStride": 3},
                        {"buffer": 0, "byteOffset": 20, "byteLength": 36, "byteStride": 6},
                        {"buffer": 0, "byteOffset": 56, "byteLength": 24}],
        "accessors": [{"bufferView": 0, "componentType": 5120, "count": 6, "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5122, "normalized": True, "count": 6, "type": "VEC3"},
                      {"bufferView": 2, "componentType": 5123, "normalized": True, "count": 6, "type": "VEC2"}],
        "materials": [{"name": "glass \\u00e9", "alphaMode": "MASK", "alphaCutoff": 0.25, "doubleSided": True,
                       "extensions": {"KHR_materials_unlit": {}}}],
        "meshes": [{"name": "tri", "primitives": [
            {"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "material": 0},
            {"attributes": {"POSITION": 0}}]}],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.5, "znear": 0.1, "zfar": 50.0, "aspectRatio": 1.5}}],
        "nodes": [
            {"name": "root", "translation": [1, 2, 3], "children": [1, 2]},
            {"name": "arm", "rotation": [0, 0.7071067811865476, 0, 0.7071067811865476], "scale": [2, 2, 2], "mesh": 0,
             "children": [3]},
            {"name": "eye", "camera": 0, "translation": [0, 0, 5]},
            {"name": "hand", "translation": [1, 0, 0], "mesh": 0},
            {"name": "unused", "mesh": 0}],
        "scenes": [{"nodes": [0]}],
    }
    text = json.dumps(doc).encode()
    text += b" " * (-len(text) % 4)
    while len(blob) % 4:
        blob.append(0)
    glb = (b"glTF" + struct.pack("<II", 2, 12 + 8 + len(text) + 8 + len(blob)) + struct.pack("<II", len(text), 0x4e4f534a) +
           text + struct.pack("<II", len(blob), 0x004e4942) + bytes(blob))
    path = tmp_path / "tree.glb"
    path.write_bytes(glb)
    loaded = R.load_scene(path)
    counts = loaded.counts()
    assert counts["geometries"] == 2 and counts["instances"] == 4 and counts["materials"] == 2 and counts["cameras"] == 1
    first, second = loaded.geometry(0), loaded.geometry(1)
    assert first["name"] == "tri.0" and second["name"] == "tri.1"
    assert first["material"] == 1 and second["material"] == 0
    assert np.array_equal(first["vertices"][:, 0:3], positions.astype(np.float32))
    want_normals = np.maximum(normals.astype(np.float32) / np.float32(32767.0), np.float32(-1.0))
    assert np.array_equal(first["vertices"][:, 3:6], want_normals)
    assert np.array_equal(first["vertices"][:, 10:12], uvs.astype(np.float32) / np.float32(65535.0))
    assert np.array_equal(second["indices"].reshape(-1), np.arange(6, dtype=np.uint32))
    material, cutoff, flags = loaded.material(1)
    assert np.float32(cutoff) == np.float32(0.25) and flags == (1 | 2 | 4)  # double sided, unlit, alpha mask
    # depth first in node order: arm (2 primitives), then hand under it (2 primitives); "unused" is not in the scene
    root = np.eye(4, dtype=np.float32)
    root[:3, 3] = [1, 2, 3]
    s = np.float32(0.7071067811865476)
    x, y, z, w = np.float32(0), s, np.float32(0), s
    rot = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 0],
                    [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w), 0],
                    [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y), 0], [0, 0, 0, 1]], dtype=np.float32)
    arm_local = rot.copy()
    arm_local[:3, :3] *= np.float32(2)
    arm = root @ arm_local
    hand_local = np.eye(4, dtype=np.float32)
    hand_local[0, 3] = 1
    hand = arm @ hand_local
    got = [loaded.instance(k) for k in range(4)]
    assert [g for g, _ in got] == [0, 1, 0, 1]
    for (_, m), want in zip(got, (arm, arm, hand, hand)):
        assert np.allclose(m.reshape(4, 4).T, want, rtol=0, atol=1e-6)
    camera = loaded.camera(0)
    assert np.allclose(camera["transform"].reshape(4, 4).T[:3, 3], [1, 2, 8])
    assert camera["farPlane"] == 50.0 and np.float32(camera["aspectRatio"]) == np.float32(1.5)
    assert camera["aperture"] == 0.0 and camera["focusDistance"] == 10.0  # the extras' defaults


def test_published_minimal_document(R, tmp_path):
    """A document this repository's writer did not make: the minimal file of the Khronos glTF tutorial
    (one triangle, 16-bit indices and positions in one embedded base64 buffer of 44 bytes), typed in as published.
    What must come out is known from the tutorial's own text: indices 0 1 2, corners (0,0,0) (1,0,0) (0,1,0), one
    node without a transform, no material (so the loader's default material 0), no camera, no light."""
    text = """{
  "scene": 0,
  "scenes" : [ { "nodes" : [ 0 ] } ],
  "nodes" : [ { "mesh" : 0 } ],
  "meshes" : [ { "primitives" : [ { "attributes" : { "POSITION" : 1 }, "indices" : 0 } ] } ],
  "buffers" : [ {
      "uri" : "data:application/octet-stream;base64,AAABAAIAAAAAAAAAAAAAAAAAAAAAAIA/AAAAAAAAAAAAAAAAAACAPwAAAAA=",
      "byteLength" : 44 } ],
  "bufferViews" : [
    { "buffer" : 0, "byteOffset" : 0, "byteLength" : 6, "target" : 34963 },
    { "buffer" : 0, "byteOffset" : 8, "byteLength" : 36, "target" : 34962 } ],
  "accessors" : [
    { "bufferView" : 0, "byteOffset" : 0, "componentType" : 5123, "count" : 3, "type" : "SCALAR",
      "max" : [ 2 ], "min" : [ 0 ] },
    { "bufferView" : 1, "byteOffset" : 0, "componentType" : 5126, "count" : 3, "type" : "VEC3",
      "max" : [ 1.0, 1.0, 0.0 ], "min" : [ 0.0, 0.0, 0.0 ] } ],
  "asset" : { "version" : "2.0" }
}"""
    path = tmp_path / "triangle.gltf"
    path.write_text(text)
    loaded = R.load_scene(path)
    counts = loaded.counts()
    assert counts["geometries"] == 1 and counts["instances"] == 1
    assert counts["materials"] == 1 and counts["cameras"] == 0 and counts["lights"] == 0
    mesh = loaded.geometry(0)
    assert np.array_equal(mesh["indices"], np.array([[0, 1, 2]], dtype=np.uint32))
    assert np.array_equal(mesh["vertices"][:, 0:3], np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32))
    assert mesh["material"] == 0
    geometry, matrix = loaded.instance(0)
    assert geometry == 0 and np.array_equal(matrix.reshape(4, 4), np.eye(4, dtype=np.float32))


def test_light_units_and_scene_extras(R, tmp_path):
    """Punctual intensities are divided by the luminous efficacy (683 unless the scene overrides it)."""
    scene = workloads.cornell_box()
    path = write_scene(scene, tmp_path / "a.gltf", "gltf", extras={"raytracer_luminous_efficacy": 0.0})
    got = R.load_scene(path).light(0)
    assert np.float32(got["intensity"]) == np.float32(np.float32(scene["lights"][0]["intensity"]) / np.float32(683.0))
    assert got["range"] == float("inf") and np.float32(got["outerCone"]) == np.float32(0.7853982)


def test_error_messages(R, tmp_path):
    def load(doc, name="bad.gltf"):
        path = tmp_path / name
        path.write_text(json.dumps(doc) if isinstance(doc, dict) else doc)
        return R.load_scene(path)

    base = {"asset": {"version": "2.0"}}
    with pytest.raises(engine.EngineError, match="Expected a glTF 2.0 .gltf or .glb scene"):
        R.load_scene(tmp_path / "scene.obj")
    with pytest.raises(engine.EngineError, match="glTF contains no scene"):
        load(base)
    with pytest.raises(engine.EngineError, match="Only triangle primitives are supported: lines"):
        load(dict(base, scenes=[{}], meshes=[{"name": "lines", "primitives": [{"mode": 1, "attributes": {}}]}]))
    with pytest.raises(engine.EngineError, match="Malformed mesh primitive"):
        load(dict(base, scenes=[{}], meshes=[{"primitives": [{"attributes": {"NORMAL": 0}}]}]))
    with pytest.raises(engine.EngineError, match="Orthographic cameras are not supported: ortho"):
        load(dict(base, scenes=[{"nodes": [0]}], nodes=[{"camera": 0}],
                  cameras=[{"name": "ortho", "type": "orthographic", "orthographic": {}}]))
    with pytest.raises(engine.EngineError, match="Required glTF extension is not supported: KHR_draco"):
        load(dict(base, scenes=[{}], extensionsRequired=["KHR_draco_mesh_compression"]))
    with pytest.raises(engine.EngineError, match="Invalid glTF JSON"):
        load('{"asset": {"version": "2.0"}, }')
    blob = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32).tobytes() + np.array([0, 1, 5], np.uint16).tobytes()
    import base64
    doc = dict(base, scenes=[{}], buffers=[{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," +
                                            base64.b64encode(blob).decode()}],
               bufferViews=[{"buffer": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 6}],
               accessors=[{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
                          {"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"}],
               meshes=[{"name": "m", "primitives": [{"attributes": {"POSITION": 0}, "indices": 1}]}])
    with pytest.raises(engine.EngineError, match="Invalid triangle indices: m.0"):
        load(doc)
    # offsets, lengths and counts that would wrap when cast or summed: a negative byteOffset, a fractional
    # one, a count whose stride * (count - 1) overflows 64 bits, an accessor offset past its view, a view
    # whose offset + length wraps
    import copy

    def broken(change):
        d = copy.deepcopy(doc)
        d["accessors"][1] = {"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"}
        change(d)
        return d

    for change, message in (
            (lambda d: d["bufferViews"][0].update(byteOffset=-1, byteLength=1), "non-negative whole number"),
            (lambda d: d["accessors"][0].update(byteOffset=-16), "non-negative whole number"),
            (lambda d: d["accessors"][0].update(byteOffset=1.5), "non-negative whole number"),
            (lambda d: d["accessors"][0].update(count=-3), "non-negative whole number"),
            (lambda d: d["accessors"][0].update(count=2 ** 70), "non-negative whole number"),
            (lambda d: d["accessors"][0].update(count=(2 ** 64) // 12 + 2), "non-negative whole number"),
            (lambda d: d["accessors"][0].update(count=2 ** 52), "too many elements"),
            (lambda d: d["accessors"][0].update(count=2 ** 31), "reaches past the end"),
            (lambda d: (d["bufferViews"][0].update(byteStride=2 ** 40), d["accessors"][0].update(count=2 ** 30)),
             "reaches past the end"),
            (lambda d: d["accessors"][0].update(byteOffset=40), "reaches past the end"),
            (lambda d: d["bufferViews"][1].update(byteOffset=2 ** 53, byteLength=6), "reaches past the end"),
            (lambda d: d["bufferViews"][1].update(byteOffset=40, byteLength=2 ** 53), "reaches past the end"),
            (lambda d: (d["accessors"][0].pop("bufferView"), d["accessors"][0].update(count=2 ** 40)), "too many elements")):
        with pytest.raises(engine.EngineError, match=message):
            load(broken(change))


def png_reference(data):
    """Independent PNG decode in numpy (8-bit, non-interlaced, colour types 0 / 2 / 6) for the comparison."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, header = 8, b"", None
    while at < len(data):
        length, kind = struct.unpack(">I4s", data[at:at + 8])
        body = data[at + 8:at + 8 + length]
        if kind == b"IHDR":
            header = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat += body
        at += 12 + length
    w, h, depth, colour, _, _, interlace = header
    assert depth == 8 and interlace == 0 and colour in (0, 2, 6)
    channels = {0: 1, 2: 3, 6: 4}[colour]
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + w * channels)
    out = np.zeros((h, w * channels), dtype=np.int32)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        up = out[y - 1] if y else np.zeros(w * channels, dtype=np.int32)
        for x in range(w * channels):
            a = out[y, x - channels] if x >= channels else 0
            b = up[x]
            c = up[x - channels] if x >= channels else 0
            if f == 0:
                pred = 0
            elif f == 1:
                pred = a
            elif f == 2:
                pred = b
            elif f == 3:
                pred = (a + b) // 2
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
            out[y, x] = (line[x] + pred) & 255
    px = out.reshape(h, w, channels).astype(np.uint8)
    rgba = np.full((h, w, 4), 255, dtype=np.uint8)
    if channels == 1:
        rgba[..., :3] = px
    else:
        rgba[..., :channels] = px
    return rgba


def test_png_reader(R):
    rng = np.random.default_rng(3)
    image = rng.integers(0, 256, (37, 23, 4), dtype=np.uint8)
    image[5:20, 3:17] = image[5, 3]  # flat region: long runs for the row filters
    for colour in (6, 2, 0):
        want = image.copy()
        if colour != 6:
            want[..., 3] = 255
        if colour == 0:
            want[..., 1] = want[..., 2] = want[..., 0]
        data = encode_png(image, colour)
        assert np.array_equal(png_reference(data), want)
        assert np.array_equal(R.decode_png(data), want)
    with pytest.raises(engine.EngineError):
        R.decode_png(b"not a png at all")
    truncated = encode_png(image)[:200]
    with pytest.raises(engine.EngineError):
        R.decode_png(truncated)
    # a header that promises 2^31 x 2^31 pixels over a few hundred bytes of data is refused, not allocated
    data = bytearray(encode_png(image))
    data[16:24] = struct.pack(">II", 1 << 31, 1 << 31)
    data[29:33] = struct.pack(">I", zlib.crc32(bytes(data[12:29])))
    with pytest.raises(engine.EngineError):
        R.decode_png(bytes(data))


def test_jpeg_reader(R, tmp_path):
    """Every kind of file the reader handles (baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0, grey, restart
    intervals, optimised tables, sizes that are no multiple of the MCU), against the pixels libjpeg-turbo
    decodes from the same bytes (tests/golden/make_jpeg_fixtures.py): the reader uses the same integer
    inverse DCT, chroma filter and colour tables, so the comparison is exact.  What stb_image, the
    reference's decoder, would give for these files is not pinned by anything the reference holds."""
    fixtures = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_fixtures.npz"))
    names = sorted(k[:-5] for k in fixtures.files if k.endswith("_file"))
    assert len(names) == 11
    for name in names:
        got = R.decode_jpeg(fixtures[name + "_file"].tobytes())
        want = fixtures[name + "_rgb"]
        assert got.shape == want.shape[:2] + (4,), name
        assert np.array_equal(got[..., :3], want), name
        assert (got[..., 3] == 255).all(), name
    for bad in (b"\xff\xd8", b"\xff\xd8\xff\xd9", b"not a jpeg at all" * 4):
        with pytest.raises(engine.EngineError):
            R.decode_jpeg(bad)
    # a JPEG texture in a scene arrives decoded, like a PNG one
    scene = workloads.cornell_box()
    data = fixtures["baseline_420_file"].tobytes()
    path = write_scene(scene, tmp_path / "jpeg.gltf", container="gltf")
    document = json.load(open(path))
    open(tmp_path / "wall.jpg", "wb").write(data)
    document["images"] = [{"uri": "wall.jpg"}]
    document["textures"] = [{"source": 0}]
    document["materials"][0]["pbrMetallicRoughness"]["baseColorTexture"] = {"index": 0}
    json.dump(document, open(path, "w"))
    loaded = R.load_scene(path)
    image = loaded.image(0)
    assert image["width"] == 61 and image["height"] == 43
    assert np.array_equal(image["rgba"][..., :3], fixtures["baseline_420_rgb"])


def test_hdr_reader(R):
    rng = np.random.default_rng(4)
    pixels = (rng.random((19, 40, 3)) ** 4 * 50.0).astype(np.float32)
    pixels[3:9, 5:30] = [2.0, 0.5, 0.25]  # runs
    pixels[0, 0] = 0.0
    rgbe = float_to_rgbe(pixels)
    want = rgbe_to_float(rgbe)
    # the format keeps 8 mantissa bits of the largest channel (truncated)
    assert (np.abs(want[..., :3] - pixels) <= pixels.max(axis=2, keepdims=True) / 128.0 + 1e-6).all()
    for rle in (True, False):
        assert np.array_equal(R.decode_hdr(encode_hdr(rgbe, rle=rle)), want)
    narrow = float_to_rgbe(pixels[:, :5])  # rows under 8 pixels are never run-length encoded
    assert np.array_equal(R.decode_hdr(encode_hdr(narrow)), rgbe_to_float(narrow))
    with pytest.raises(engine.EngineError):
        R.decode_hdr(b"#?RADIANCE\nFORMAT=32-bit_rle_xyze\n\n-Y 1 +X 1\n\0\0\0\0")
    with pytest.raises(engine.EngineError):  # more pixels promised than the file can hold
        R.decode_hdr(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2000000000 +X 2000000000\n" + b"\2\2\0\10" * 16)


ASSETS = "/root/reference/assets"


@pytest.mark.skipif(not os.path.isdir(ASSETS), reason="the reference's asset files are not on this machine")
def test_readers_on_the_reference_assets(R):
    """The image files the reference ships (inputs only; nothing of them is copied into the repository)."""
    for name in ("checker_blue.png", "checker_green.png", "checker_red.png"):
        data = open(os.path.join(ASSETS, "textures", name), "rb").read()
        got = R.decode_png(data)
        assert got.shape == (512, 512, 4)
        header = struct.unpack(">IIBBBBB", data[16:29])
        if header[2] == 8 and header[3] in (0, 2, 6) and header[6] == 0:
            small = png_reference(data)[:16]
            assert np.array_equal(got[:16], small)
    for name, shape in (("rtiow_sky.hdr", (512, 1024)),):
        data = open(os.path.join(ASSETS, "hdri", name), "rb").read()
        got = R.decode_hdr(data)
        assert got.shape[:2] == shape and np.isfinite(got).all() and (got[..., 3] == 1.0).all() and got[..., :3].max() > 1.0
    # the progressive JPEG pictures of the reference's readme, against libjpeg-turbo where Pillow is installed
    pictures = os.path.join(os.path.dirname(ASSETS), "img", "github")
    for name in ("dragon_cornell.jpg", "sphere_field.jpg"):
        path = os.path.join(pictures, name)
        if not os.path.isfile(path):
            continue
        data = open(path, "rb").read()
        got = R.decode_jpeg(data)
        assert got.shape == (720, 1280, 4) and (got[..., 3] == 255).all()
        try:
            import io
            from PIL import Image
        except ImportError:
            continue
        assert np.array_equal(got[..., :3], np.asarray(Image.open(io.BytesIO(data)).convert("RGB")))


def read_exr(path):
    """Minimal reader for the scan-line, uncompressed, half-float files rt::writeExr produces."""
    data = open(path, "rb").read()
    assert struct.unpack("<II", data[:8]) == (20000630, 2)
    at, channels, window = 8, [], None
    while data[at] != 0:
        end = data.index(b"\0", at)
        name = data[at:end].decode()
        at = end + 1
        end = data.index(b"\0", at)
        kind = data[at:end].decode()
        at = end + 1
        size, = struct.unpack("<I", data[at:at + 4])
        body = data[at + 4:at + 4 + size]
        at += 4 + size
        if name == "channels":
            p = 0
            while body[p] != 0:
                e = body.index(b"\0", p)
                channels.append((body[p:e].decode(), struct.unpack("<I", body[e + 1:e + 5])[0]))
                p = e + 1 + 16
        elif name == "dataWindow":
            window = struct.unpack("<4i", body)
        elif name == "compression":
            assert body == b"\0" and kind == "compression"
        elif name == "lineOrder":
            assert body == b"\0"
    at += 1
    assert channels == [("A", 1), ("B", 1), ("G", 1), ("R", 1)]
    w, h = window[2] + 1, window[3] + 1
    offsets = struct.unpack(f"<{h}Q", data[at:at + 8 * h])
    image = np.zeros((h, w, 4), dtype=np.float32)
    for y in range(h):
        row, size = struct.unpack("<iI", data[offsets[y]:offsets[y] + 8])
        assert row == y and size == w * 8
        line = np.frombuffer(data, dtype=np.float16, count=4 * w, offset=offsets[y] + 8).reshape(4, w).astype(np.float32)
        image[y, :, 3], image[y, :, 2], image[y, :, 1], image[y, :, 0] = line[0], line[1], line[2], line[3]
    return image


def test_image_writers(R, tmp_path):
    """writePng / writeExr flip the renderer's bottom-up rows; PNG is lossless, EXR rounds to binary16."""
    rng = np.random.default_rng(8)
    rgba = rng.integers(0, 256, (13, 21, 4), dtype=np.uint8)
    R.write_png(tmp_path / "a.png", rgba)
    data = (tmp_path / "a.png").read_bytes()
    assert np.array_equal(png_reference(data), rgba[::-1])
    assert np.array_equal(R.decode_png(data), rgba[::-1])
    linear = (rng.random((13, 21, 4)) ** 3 * 40.0).astype(np.float32)
    linear[0, 0] = [0.0, 1e-9, 70000.0, 1.0]   # zero, underflow, overflow to infinity
    linear[0, 1] = [6.1e-5, 3e-6, 65504.0, -2.5]  # smallest normal, a subnormal, largest finite, negative
    R.write_exr(tmp_path / "a.exr", linear)
    with np.errstate(over="ignore"):
        want = linear[::-1].astype(np.float16).astype(np.float32)  # numpy rounds to nearest even as well
    assert np.array_equal(read_exr(tmp_path / "a.exr"), want)
    with pytest.raises(engine.EngineError, match="Unable to write PNG"):
        R.write_png(tmp_path / "missing" / "a.png", rgba)


def test_command_line_options(R):
    """Option handling of the command-line renderer, as far as it goes without a GPU."""
    import subprocess
    if not os.path.exists(R.CLI_PATH):
        pytest.skip("the raytracer executable is not built")

    def run(*args):
        p = subprocess.run([R.CLI_PATH, *args], capture_output=True, text=True, timeout=60)
        return p.returncode, p.stdout, p.stderr

    code, _, err = run()
    assert code == 1 and err.startswith("error: usage: raytracer <scene.gltf|scene.glb>")
    assert run("x.glb", "--samples")[2] == "error: Missing option value\n"
    assert run("x.glb", "--samples", "0")[2] == "error: Invalid sample count\n"
    assert run("x.glb", "--width", "12a")[2] == "error: Invalid width\n"
    assert run("x.glb", "--profile", "draft")[2] == "error: Expected preview or final profile\n"
    assert run("x.glb", "--denoise", "maybe")[2] == "error: Expected on or off for denoise\n"
    assert run("x.glb", "--fast", "1")[2] == "error: Unknown option: --fast\n"
    assert run("x.glb", "--output", "a.jpg")[2] == "error: Output must use .png or .exr\n"
    assert "no interactive viewer" in run("x.glb")[2]
    assert "Expected a glTF 2.0 .gltf or .glb scene" in run("x.obj", "--output", "a.png")[2]
