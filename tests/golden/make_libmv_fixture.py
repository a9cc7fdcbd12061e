#!/usr/bin/env python3
"""Generates tests/golden/libmv_problems.npz from the three bundle-adjustment problems the reference ships
(/root/reference/data/libmv-ba-problems/problem_0{1,2,3}.bin, "final camera motion refinement" steps of Tears of Steel scenes; reader:
/root/reference/examples/libmv_bundle_adjuster.cc:376-455 — endian flag, marker space flag, 8 float intrinsics, cameras {image, R column-major,
t}, points {track, X}, markers {image, track, x, y}).  Kept: everything but nothing derived — the visibility graph (marker image / track),
the marker coordinates, camera poses, points and intrinsics, so that tests and tools on the GPU box (which has no /root/reference) see
real sparsity and real geometry.  Run in the build container only."""
import struct, sys, os
import numpy as np

SRC = "/root/reference/data/libmv-ba-problems/problem_0%d.bin"
out = {}
for k in (1, 2, 3):
    b = open(SRC % k, "rb").read()
    end = "<" if b[0:1] == b"v" else ">"
    off = 1
    out[f"p{k}_image_space"] = np.array([1 if b[off:off + 1] == b"P" else 0], np.int32); off += 1
    out[f"p{k}_intrinsics"] = np.array(struct.unpack_from(end + "8f", b, off), np.float32); off += 32
    nc, = struct.unpack_from(end + "i", b, off); off += 4
    cam = np.frombuffer(b, dtype=np.dtype([("image", end + "i4"), ("R", end + "f4", (9,)), ("t", end + "f4", (3,))]), count=nc, offset=off); off += nc * 52
    npt, = struct.unpack_from(end + "i", b, off); off += 4
    pts = np.frombuffer(b, dtype=np.dtype([("track", end + "i4"), ("X", end + "f4", (3,))]), count=npt, offset=off); off += npt * 16
    nm, = struct.unpack_from(end + "i", b, off); off += 4
    mk = np.frombuffer(b, dtype=np.dtype([("image", end + "i4"), ("track", end + "i4"), ("x", end + "f4"), ("y", end + "f4")]), count=nm, offset=off); off += nm * 16
    assert off == len(b), (off, len(b))
    out[f"p{k}_camera_image"] = cam["image"].astype(np.int32)
    out[f"p{k}_camera_R"] = cam["R"].astype(np.float32)       # column-major 3x3, as stored
    out[f"p{k}_camera_t"] = cam["t"].astype(np.float32)
    out[f"p{k}_point_track"] = pts["track"].astype(np.int32)
    out[f"p{k}_point_X"] = pts["X"].astype(np.float32)
    out[f"p{k}_marker_image"] = mk["image"].astype(np.int32)
    out[f"p{k}_marker_track"] = mk["track"].astype(np.int32)
    out[f"p{k}_marker_xy"] = np.stack([mk["x"], mk["y"]], 1).astype(np.float32)
    print(f"problem_0{k}: {nc} cameras, {npt} points, {nm} markers")
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmv_problems.npz")
np.savez_compressed(dst, **out)
print(dst, os.path.getsize(dst), "bytes")
