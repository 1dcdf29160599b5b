"""Texture ingestion: how the reference flattens Bevy `Image`s into the `textures` / `texture_data` buffers the
path reads (reference src/data.rs:53-56 `TextureData`, :99-160 `TextureData::append_texture`, called for every image
asset by `extract_textures`, src/extract.rs:257-295).

The five formats are restated with the reference's arithmetic, quirks included, so that images fed through this
module produce the bytes the Rust host would upload:
  Rgba8UnormSrgb  every byte * (1/255) in f32, no sRGB decode                                   -> 4 channels
  Rgba16Float     every half widened to f32, then ALSO * (1/255)                                 -> 4 channels
  Rgb9e5Ufloat    per 4 bytes, the reference's own u8 bit arithmetic builds three little-endian
                  half bit patterns (most of its cross-byte terms shift to zero)                 -> 3 channels
  R8Unorm         every byte * (1/255)                                                           -> 1 channel
  Rg8Unorm        every byte * (1/255)                                                           -> 2 channels
Any other format makes the reference panic ("Texture format ... is not supported."); here it raises ValueError.
"""
from __future__ import annotations

import numpy as np

from .abi import TEXTURE_DT

DIV_255 = np.float32(1.0) / np.float32(255.0)   # `const DIV_255: f32 = 1.0 / 255.0;` (src/data.rs:102)

FORMATS = ("Rgba8UnormSrgb", "Rgba16Float", "Rgb9e5Ufloat", "R8Unorm", "Rg8Unorm")


def _rgb9e5_as_the_reference_reads_it(d: np.ndarray) -> np.ndarray:
    """src/data.rs:120-135. `d` is (n, 4) uint8; all arithmetic is u8 (shifts drop the bits that leave the byte)."""
    d = d.astype(np.uint8)
    d0, d1, d2, d3 = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
    u8 = np.uint8
    e = (d3 << u8(2)) & u8(0b011111)
    r_lo = e | (d0 >> u8(7))
    r_hi = (d0 << u8(1)) | ((d1 & u8(0b1)) >> u8(7))
    g_lo = e | ((d1 & u8(0b01)) >> u8(6))
    g_hi = (d1 << u8(2)) | ((d2 & u8(0b11)) >> u8(6))
    b_lo = e | ((d2 & u8(0b001)) >> u8(5))
    b_hi = (d2 << u8(3)) | ((d3 & u8(0b111)) >> u8(5))
    out = np.empty((d.shape[0], 3), dtype=np.float32)
    for k, (lo, hi) in enumerate(((r_lo, r_hi), (g_lo, g_hi), (b_lo, b_hi))):
        bits = lo.astype(np.uint16) | (hi.astype(np.uint16) << np.uint16(8))   # f16::from_le_bytes([lo, hi])
        out[:, k] = bits.view(np.float16).astype(np.float32)
    return out


class TextureData:
    """`data`: the f32 texel pool; `textures`: one descriptor per appended image."""

    def __init__(self):
        self._chunks: list[np.ndarray] = []
        self._len = 0
        self.textures: list[np.void] = []

    @property
    def data(self) -> np.ndarray:
        return np.concatenate(self._chunks).astype(np.float32) if self._chunks else np.zeros(0, dtype=np.float32)

    def append_texture(self, image_data: bytes | np.ndarray, width: int, height: int, format: str):
        """Flatten one image (`image.data`, `image.width()`, `image.height()`, `image.texture_descriptor.format`)."""
        raw = np.frombuffer(bytes(image_data), dtype=np.uint8) if not isinstance(image_data, np.ndarray) else image_data.astype(np.uint8).reshape(-1)
        offset = self._len
        if format == "Rgba8UnormSrgb":
            texels, channels = raw.astype(np.float32) * DIV_255, 4
        elif format == "Rgba16Float":
            halves = raw[: raw.size // 2 * 2].view("<f2")            # chunks(2): a trailing odd byte would panic in the reference
            if raw.size % 2:
                raise ValueError("Rgba16Float image data has an odd number of bytes")
            texels, channels = halves.astype(np.float32) * DIV_255, 4
        elif format == "Rgb9e5Ufloat":
            if raw.size % 4:
                raise ValueError("Rgb9e5Ufloat image data is not a multiple of 4 bytes")
            texels, channels = _rgb9e5_as_the_reference_reads_it(raw.reshape(-1, 4)).reshape(-1), 3
        elif format == "R8Unorm":
            texels, channels = raw.astype(np.float32) * DIV_255, 1
        elif format == "Rg8Unorm":
            texels, channels = raw.astype(np.float32) * DIV_255, 2
        else:
            raise ValueError(f"Texture format {format} is not supported.")
        texels = texels.astype(np.float32)
        self._chunks.append(texels)
        self._len += texels.size
        t = np.zeros((), dtype=TEXTURE_DT)
        t["width"], t["height"], t["offset"], t["format"] = width, height, offset, channels
        self.textures.append(t)
        return t

    def texture_array(self) -> np.ndarray:
        return np.array(self.textures, dtype=TEXTURE_DT) if self.textures else np.zeros(0, dtype=TEXTURE_DT)
