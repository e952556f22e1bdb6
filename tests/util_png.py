"""Reading the reference's rendered micro-frontend output (tests/golden/tutorial_cell13.png).

Test infrastructure: a dependency-free PNG decoder (8-bit RGB / RGBA, non-interlaced: what matplotlib's Agg
backend writes) and the geometry recovery of `imshow` panels -- the axes rectangles and the cell grid are
DETECTED from the pixels, nothing about the figure's layout is assumed beyond "N panels side by side, each a
nearest-neighbour rendering of a [rows, cols] array through a known byte colormap".
"""
import json
import os
import struct
import zlib

import numpy as np


def decode_png(data):
    """bytes -> uint8 [H, W, C] (C = 3 or 4)."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(kind + body) == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0], "PNG chunk CRC"
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        pos += 12 + n
    W, H, depth, ctype, _, _, interlace = hdr
    assert depth == 8 and ctype in (2, 6) and interlace == 0, hdr
    C = 3 if ctype == 2 else 4
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(H, 1 + W * C)
    out = np.zeros((H, W * C), dtype=np.int32)
    zero = np.zeros(W * C, dtype=np.int32)
    for y in range(H):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        up = out[y - 1] if y else zero
        if f == 0:
            out[y] = line
        elif f == 2:
            out[y] = (line + up) & 255
        else:                                        # Sub / Average / Paeth: left-neighbour recurrences
            row = out[y]
            for x in range(W * C):
                a = row[x - C] if x >= C else 0
                b = up[x]
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + b) >> 1
                else:
                    c = up[x - C] if x >= C else 0
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                row[x] = (line[x] + p) & 255
    return out.reshape(H, W, C).astype(np.uint8)


def load_viridis(golden_dir):
    return np.array(json.load(open(os.path.join(golden_dir, "viridis_bytes.json")))["rgb"], dtype=np.int64)


def _runs(flags):
    idx = np.nonzero(flags)[0]
    if idx.size == 0:
        return []
    cut = np.nonzero(np.diff(idx) > 1)[0] + 1
    return [(int(r[0]), int(r[-1]) + 1) for r in np.split(idx, cut)]


def _grid(bounds, ncell, extent):
    """Detected cell boundaries (pixel offsets inside the panel) -> the ncell + 1 boundaries of a UNIFORM grid.

    The detected boundaries must be consecutive ones starting with the first (every gap = the step rounded down
    or up); the grid is their least-squares line, extended to ncell cells; it has to end where the panel ends.
    """
    b = np.asarray(bounds, dtype=np.float64)
    assert b.size >= 8, "too few cell boundaries visible"
    step0 = (b[-1] - b[0]) / (b.size - 1)
    gaps = np.diff(b)
    assert set(np.unique(gaps)) <= {np.floor(step0), np.ceil(step0)}, "visible boundaries are not consecutive"
    assert b[0] <= np.ceil(step0), "first visible boundary is not the first cell's"
    k = np.arange(1, b.size + 1)
    s, o = np.polyfit(k, b, 1)
    assert np.abs(o + s * k - b).max() < 0.5 + 1e-9, "cell grid is not uniform"
    fit = o + s * np.arange(ncell + 1)
    assert fit[ncell - 1] < extent - 1 and fit[ncell] > extent - 1 - s / 2, ("grid does not end with the panel", fit[-2:], extent)
    return fit


def imshow_panels(img, lut, shape):
    """RGB(A) image holding imshow panels of [rows, cols] arrays -> list of [rows, cols, 3] cell colours.

    Panels = the maximal rectangles made of colormap colours only.  Inside a panel the row / column boundaries
    are wherever two neighbouring pixel rows / columns differ; cells are read as the block strictly inside
    their boundaries, which must be one colour.
    """
    rows, cols = shape
    rgb = img[..., :3].astype(np.int64)
    key = (rgb[..., 0] << 16) | (rgb[..., 1] << 8) | rgb[..., 2]
    lkey = (lut[:, 0] << 16) | (lut[:, 1] << 8) | lut[:, 2]
    inmap = np.isin(key, lkey)
    H, W = inmap.shape
    yr = _runs(inmap.sum(1) > W // 2)
    assert len(yr) == 1, yr
    y0, y1 = yr[0]
    out = []
    for x0, x1 in _runs(inmap[y0:y1].sum(0) > (y1 - y0) // 2):
        assert inmap[y0:y1, x0:x1].all(), "panel holds pixels that are not colormap colours"
        sub = key[y0:y1, x0:x1]
        rb = np.nonzero((sub[1:] != sub[:-1]).any(1))[0] + 1
        cb = np.nonzero((sub[:, 1:] != sub[:, :-1]).any(0))[0] + 1
        fr, fc = _grid(rb, rows, y1 - y0), _grid(cb, cols, x1 - x0)
        cells = np.zeros((rows, cols, 3), dtype=np.int64)
        for i in range(rows):
            a, b = max(0, int(np.ceil(fr[i] + 0.5))), min(y1 - y0, int(np.floor(fr[i + 1] - 0.5)) + 1)
            for j in range(cols):
                c, d = max(0, int(np.ceil(fc[j] + 0.5))), min(x1 - x0, int(np.floor(fc[j + 1] - 0.5)) + 1)
                blk = sub[a:b, c:d]
                assert blk.size and (blk == blk.flat[0]).all(), ("cell is not one colour", i, j)
                cells[i, j] = rgb[y0 + a, x0 + c]
        out.append(cells)
    return out


def imshow_expected_index(x):
    """matplotlib `imshow(x)` with its defaults: Normalize(min, max) then the 256-entry colormap:
    index = floor(256 * (x - min) / (max - min)), the top value clipped to 255."""
    x = np.asarray(x, dtype=np.float64)
    lo, hi = x.min(), x.max()
    return np.clip(np.floor(256.0 * (x - lo) / (hi - lo)), 0, 255).astype(np.int64)


def index_distance(cells, lut, expected_idx):
    """Per cell: the smallest |i - expected| over the colormap entries i that have the cell's colour
    (two pairs of neighbouring viridis entries share a byte triple)."""
    d = np.full(expected_idx.shape, 256, dtype=np.int64)
    for i in range(lut.shape[0]):
        hit = (cells == lut[i]).all(-1)
        d = np.where(hit, np.minimum(d, np.abs(expected_idx - i)), d)
    return d
