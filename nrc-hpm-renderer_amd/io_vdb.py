"""Minimal OpenVDB (.vdb, file version >= 222, FloatGrid Tree_float_5_4_3) reader -> dense numpy volume.

Replaces `Texture3D::FromVDB` (src/Texture3D.cpp:12-82), which uses OpenVDB v10.0.0 (absent submodule):
dense-ify the grid over `file_bbox`, fill active tiles, require max == 1 (:74).
Handles what the reference's cloud files use: no zip/blosc, optional active-mask compression.
Format notes: SURVEY.md App. E.
"""
import struct
import zlib

import numpy as np


class _R:
    def __init__(self, data):
        self.d, self.p = data, 0

    def take(self, n):
        b = self.d[self.p:self.p + n]
        self.p += n
        return b

    def u8(self):
        return self.take(1)[0]

    def i8(self):
        return struct.unpack("<b", self.take(1))[0]

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def i32(self):
        return struct.unpack("<i", self.take(4))[0]

    def i64(self):
        return struct.unpack("<q", self.take(8))[0]

    def f32(self):
        return struct.unpack("<f", self.take(4))[0]

    def string(self):
        return self.take(self.u32()).decode("latin1")

    def meta(self):
        out = {}
        for _ in range(self.u32()):
            name, typ = self.string(), self.string()
            size = self.u32()
            payload = self.take(size)
            if typ == "vec3i":
                out[name] = struct.unpack("<3i", payload)
            elif typ == "int64":
                out[name] = struct.unpack("<q", payload)[0]
            elif typ == "string":
                out[name] = payload.decode("latin1")
            else:
                out[name] = payload
        return out


def _mask(r, nbits):
    return np.unpackbits(np.frombuffer(r.take(nbits // 8), np.uint8), bitorder="little").astype(bool)


def _values(r, count, mask, flags, background):
    """'compressed value array' (SURVEY App. E item 5)."""
    meta = r.i8()
    inactive0, inactive1 = background, -background if meta != 0 else background
    if meta in (2, 4, 5):
        inactive0 = r.f32()
        if meta == 5:
            inactive1 = r.f32()
    if meta == 1:
        inactive0 = -background
    sel = None
    if meta in (3, 4, 5):
        sel = _mask(r, count)
    n = int(mask.sum()) if (flags & 2) and meta != 6 else count
    if flags & 1:
        nbytes = r.i64()
        raw = zlib.decompress(r.take(nbytes)) if nbytes > 0 else r.take(-nbytes)
    else:
        raw = r.take(4 * n)
    vals = np.frombuffer(raw, "<f4", count=n)
    if n == count:
        return vals.copy()
    out = np.full(count, inactive0, np.float32)
    if sel is not None:
        out[sel] = inactive1
    out[mask] = vals
    return out


def read_vdb_dense(path):
    """Returns (volume float32 indexed [x][y][z] over file_bbox, info dict)."""
    with open(path, "rb") as f:
        r = _R(f.read())
    magic = r.i64()
    assert magic == 0x56444220, "not a VDB file"
    version = r.u32()
    r.u32()
    r.u32()                 # library major / minor
    r.u8()                  # has grid offsets
    r.take(36)              # uuid
    r.meta()
    n_grids = r.u32()
    assert n_grids >= 1
    name, gtype = r.string(), r.string()
    r.string()              # instance parent
    grid_pos, block_pos, end_pos = r.i64(), r.i64(), r.i64()
    assert "Tree_float_5_4_3" in gtype, gtype
    r.p = grid_pos
    flags = r.u32()
    gmeta = r.meta()
    bmin, bmax = np.array(gmeta["file_bbox_min"]), np.array(gmeta["file_bbox_max"])
    ext = bmax - bmin + 1
    vol = np.zeros(tuple(ext), np.float32)
    r.string()              # transform type (UniformScaleMap etc.); payload skipped by seeking via topology parse below
    # transform payload length depends on the map type; topology starts right after it.  All maps used by the
    # WDAS cloud files are UniformScaleMap = 5 x vec3d
    r.take(120)
    assert r.u32() == 1     # buffer count
    background = r.f32()
    n_tiles, n_children = r.u32(), r.u32()
    leaves = []             # (origin, value mask) in topology order
    active_voxels = 0

    def fill(o, dim, value):
        nonlocal active_voxels
        lo = np.maximum(o - bmin, 0)
        hi = np.minimum(o - bmin + dim, ext)
        if (hi > lo).all():
            vol[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = value
        active_voxels += dim ** 3

    for _ in range(n_tiles):
        o = np.array([r.i32(), r.i32(), r.i32()])
        v, act = r.f32(), r.u8()
        if act:
            fill(o, 4096, v)
    for _ in range(n_children):
        o5 = np.array([r.i32(), r.i32(), r.i32()])
        cm5, vm5 = _mask(r, 32768), _mask(r, 32768)
        vals5 = _values(r, 32768, vm5, flags, background)
        for n in np.nonzero(vm5 & ~cm5)[0]:
            fill(o5 + 128 * np.array([n >> 10, (n >> 5) & 31, n & 31]), 128, vals5[n])
        for n in np.nonzero(cm5)[0]:
            o4 = o5 + 128 * np.array([n >> 10, (n >> 5) & 31, n & 31])
            cm4, vm4 = _mask(r, 4096), _mask(r, 4096)
            vals4 = _values(r, 4096, vm4, flags, background)
            for k in np.nonzero(vm4 & ~cm4)[0]:
                fill(o4 + 8 * np.array([k >> 8, (k >> 4) & 15, k & 15]), 8, vals4[k])
            for k in np.nonzero(cm4)[0]:
                o3 = o4 + 8 * np.array([k >> 8, (k >> 4) & 15, k & 15])
                leaves.append((o3, _mask(r, 512)))
    assert r.p == block_pos, (r.p, block_pos)
    for o3, _vm in leaves:
        vm = _mask(r, 512)
        vals = _values(r, 512, vm, flags, background)
        active_voxels += int(vm.sum())
        blk = np.where(vm, vals, 0.0).astype(np.float32).reshape(8, 8, 8)   # n = x<<6 | y<<3 | z
        lo = o3 - bmin
        if (lo >= 0).all() and (lo + 8 <= ext).all():
            sub = vol[lo[0]:lo[0] + 8, lo[1]:lo[1] + 8, lo[2]:lo[2] + 8]
            sub[vm.reshape(8, 8, 8)] = blk[vm.reshape(8, 8, 8)]
        else:
            for n in np.nonzero(vm)[0]:
                p = lo + np.array([n >> 6, (n >> 3) & 7, n & 7])
                if (p >= 0).all() and (p < ext).all():
                    vol[tuple(p)] = vals[n]
    assert r.p == end_pos, (r.p, end_pos)
    info = dict(version=version, name=name, bbox_min=tuple(bmin), bbox_max=tuple(bmax), extent=tuple(int(e) for e in ext),
                file_voxel_count=gmeta.get("file_voxel_count"), active_voxels=active_voxels, flags=flags)
    return vol, info


def from_vdb(path):
    """Texture3D::FromVDB semantics: dense volume + the max==1 normalisation check (src/Texture3D.cpp:74)."""
    vol, info = read_vdb_dense(path)
    mx = float(vol.max())
    if mx != 0.0 and mx != 1.0:
        raise RuntimeError("SkyRenderer ERROR: VDB is not normalized")
    return vol, info
