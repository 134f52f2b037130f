"""Minimal OpenEXR scan-line reader/writer (FLOAT channels, NONE/ZIPS/ZIP compression).

Replaces the tinyexr calls of the reference: `ExportOutputImageToFile` (src/NrcHpmRenderer.cu:437-493) writes
RGBA FLOAT; `Reference` loads reference/<scene>/0.exr (src/Reference.cpp:608-631).  Format notes: SURVEY App. E.
"""
import struct
import zlib

import numpy as np


def _unpredict(buf):
    b = np.frombuffer(buf, np.uint8).astype(np.int32)
    # predictor: b[i] = b[i-1] + b[i] - 128 (mod 256)  -> cumulative sum
    b[1:] -= 128
    b = np.cumsum(b) & 0xFF
    b = b.astype(np.uint8)
    half = (len(b) + 1) // 2
    out = np.empty(len(b), np.uint8)
    out[0::2] = b[:half]
    out[1::2] = b[half:]
    return out.tobytes()


def _predict(raw):
    b = np.frombuffer(raw, np.uint8)
    half = (len(b) + 1) // 2
    t = np.empty(len(b), np.uint8)
    t[:half] = b[0::2]
    t[half:] = b[1::2]
    d = t.astype(np.int32)
    out = d.copy()
    out[1:] = (d[1:] - d[:-1] + 128) & 0xFF
    return out.astype(np.uint8).tobytes()


def read_exr(path):
    """Returns float32 array [H][W][4] in R,G,B,A order (missing channels = 0, A defaults 1)."""
    with open(path, "rb") as f:
        d = f.read()
    assert struct.unpack("<I", d[:4])[0] == 20000630, "not an EXR file"
    p = 8
    attrs = {}
    while d[p] != 0:
        e = d.index(b"\0", p)
        name = d[p:e].decode()
        p = e + 1
        e = d.index(b"\0", p)
        typ = d[p:e].decode()
        p = e + 1
        size = struct.unpack("<i", d[p:p + 4])[0]
        p += 4
        attrs[name] = (typ, d[p:p + size])
        p += size
    p += 1
    chans = []
    cb = attrs["channels"][1]
    q = 0
    while cb[q] != 0:
        e = cb.index(b"\0", q)
        cname = cb[q:e].decode()
        q = e + 1
        ptype = struct.unpack("<i", cb[q:q + 4])[0]
        q += 16
        assert ptype == 2, "only FLOAT channels supported"
        chans.append(cname)
    comp = attrs["compression"][1][0]
    xmin, ymin, xmax, ymax = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = xmax - xmin + 1, ymax - ymin + 1
    lines = {0: 1, 2: 1, 3: 16}[comp]
    n_chunks = (H + lines - 1) // lines
    offs = struct.unpack("<%dQ" % n_chunks, d[p:p + 8 * n_chunks])
    img = np.zeros((H, W, 4), np.float32)
    img[..., 3] = 1.0
    cidx = {"R": 0, "G": 1, "B": 2, "A": 3}
    for o in offs:
        y, size = struct.unpack("<ii", d[o:o + 8])
        payload = d[o + 8:o + 8 + size]
        nl = min(lines, ymax + 1 - y)
        raw_size = nl * W * 4 * len(chans)
        raw = payload if (comp == 0 or size == raw_size) else _unpredict(zlib.decompress(payload))
        a = np.frombuffer(raw, "<f4").reshape(nl, len(chans), W)
        for ci, cn in enumerate(chans):
            if cn in cidx:
                img[y - ymin:y - ymin + nl, :, cidx[cn]] = a[:, ci, :]
    return img


def write_exr(path, img, compression="zip"):
    """Write [H][W][4] float32 (RGBA) as a scan-line EXR with FLOAT channels A,B,G,R."""
    img = np.ascontiguousarray(img, np.float32)
    H, W = img.shape[:2]
    comp = {"none": 0, "zip": 3}[compression]
    lines = 16 if comp == 3 else 1

    def attr(name, typ, payload):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload

    ch = b""
    for cn in ("A", "B", "G", "R"):
        ch += cn.encode() + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1)
    ch += b"\0"
    box = struct.pack("<4i", 0, 0, W - 1, H - 1)
    hdr = struct.pack("<II", 20000630, 2)
    hdr += attr("channels", "chlist", ch)
    hdr += attr("compression", "compression", bytes([comp]))
    hdr += attr("dataWindow", "box2i", box)
    hdr += attr("displayWindow", "box2i", box)
    hdr += attr("lineOrder", "lineOrder", b"\0")
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
    hdr += attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    hdr += b"\0"
    n_chunks = (H + lines - 1) // lines
    chunks = []
    order = [3, 2, 1, 0]     # A, B, G, R
    for c in range(n_chunks):
        y0 = c * lines
        nl = min(lines, H - y0)
        a = img[y0:y0 + nl][:, :, order].transpose(0, 2, 1)      # [line][chan][x]
        raw = np.ascontiguousarray(a, "<f4").tobytes()
        payload = raw
        if comp == 3:
            z = zlib.compress(_predict(raw))
            payload = z if len(z) < len(raw) else raw
        chunks.append(struct.pack("<ii", y0, len(payload)) + payload)
    off = len(hdr) + 8 * n_chunks
    table = b""
    for ck in chunks:
        table += struct.pack("<Q", off)
        off += len(ck)
    with open(path, "wb") as f:
        f.write(hdr + table + b"".join(chunks))
