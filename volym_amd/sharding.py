"""Host-side description of the screen-tile sharding, mirrored from
volym_amd/csrc/raymarch.hip so that the multi-rank protocol can be exercised without a GPU.

Frame = tiles_x x tiles_y tiles of 16x16 pixels, tile k = ty*tiles_x + tx.  Rank r of `world`
owns the tiles k with k % world == r; its shard holds them in order of k (local index k // world),
1024 bytes each: four 8x8 sub-tiles (sub = (py>=8)*2 + (px>=8)), 64 RGBA8 pixels per sub-tile in
row-major order.  Shards are padded to ceil(n_tiles / world) tiles so every rank sends the same
number of bytes (all_gather).  Pixels outside the frame are zero.
"""
import numpy as np


def tiling(W, H):
    tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
    return tiles_x, tiles_y, tiles_x * tiles_y


def local_tiles(rank, world, n_tiles):
    return (n_tiles - rank + world - 1) // world if n_tiles > rank else 0


def shard_tiles(world, n_tiles):
    return (n_tiles + world - 1) // world


def shard_bytes(W, H, world):
    return shard_tiles(world, tiling(W, H)[2]) * 1024


def _tile_pixel_index(W, H):
    """[tile, 256] -> flat pixel index into the H*W raster, or -1 outside the frame."""
    tiles_x, tiles_y, n_tiles = tiling(W, H)
    t = np.arange(256)
    sub, lane = t // 64, t % 64
    px = (sub & 1) * 8 + (lane & 7)
    py = (sub >> 1) * 8 + (lane >> 3)
    k = np.arange(n_tiles)
    gx = (k % tiles_x)[:, None] * 16 + px[None, :]
    gy = (k // tiles_x)[:, None] * 16 + py[None, :]
    idx = gy * W + gx
    idx[(gx >= W) | (gy >= H)] = -1
    return idx


def pack_shard(frame_rgba8, rank, world):
    """Raster [H, W, 4] uint8 -> this rank's shard bytes (what the kernel writes)."""
    H, W, _ = frame_rgba8.shape
    n_tiles = tiling(W, H)[2]
    idx = _tile_pixel_index(W, H)
    flat = np.concatenate([frame_rgba8.reshape(-1, 4), np.zeros((1, 4), np.uint8)])   # index -1 -> zeros
    out = np.zeros((shard_tiles(world, n_tiles), 256, 4), np.uint8)
    mine = np.arange(rank, n_tiles, world)
    out[: mine.size] = flat[idx[mine]]
    return out.reshape(-1)


def assemble(gathered, W, H, world):
    """world shards back to back (rank order) -> raster [H, W, 4] uint8 (volym_assemble)."""
    n_tiles = tiling(W, H)[2]
    st = shard_tiles(world, n_tiles)
    g = np.asarray(gathered, np.uint8).reshape(world, st, 256, 4)
    idx = _tile_pixel_index(W, H)
    k = np.arange(n_tiles)
    tiles = g[k % world, k // world]                      # [n_tiles, 256, 4]
    frame = np.zeros((H * W, 4), np.uint8)
    ok = idx >= 0
    frame[idx[ok]] = tiles[ok]
    return frame.reshape(H, W, 4)
