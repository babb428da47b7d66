"""Image output for the headless path (replaces the swapchain blit, reference rvpt.cpp:911-964 / tex_sample.frag):
PFM keeps the float radiance, PNG stores the reference's display format (clamped UNORM8)."""
from __future__ import annotations

import struct
import zlib

import numpy as np


def write_pfm(path, rgb: np.ndarray) -> None:
    """Portable float map, little-endian, 3 channels, bottom row first (PFM convention)."""
    img = np.ascontiguousarray(rgb[..., :3], dtype="<f4")
    h, w = img.shape[:2]
    with open(path, "wb") as f:
        f.write(f"PF\n{w} {h}\n-1.0\n".encode())
        f.write(img[::-1].tobytes())


def read_pfm(path) -> np.ndarray:
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return data[::-1].copy()


def write_png(path, rgba8: np.ndarray) -> None:
    """Minimal PNG encoder (8-bit RGB, no dependencies)."""
    img = np.ascontiguousarray(rgba8[..., :3], dtype=np.uint8)
    h, w = img.shape[:2]
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
