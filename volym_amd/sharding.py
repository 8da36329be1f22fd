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


# ---- packed shards (volym_pack_shard / volym_assemble_packed, raymarch_kernels.h) ------------------------------------------
# A packed shard is [header: one (uint32, uint32) pair per shard tile, padded to 1 KiB][1 KiB tiles].  header[t] = (slot, 0)
# for a tile stored at tiles[slot], or (0xffffffff, value) for a tile whose 256 pixels all equal `value` (not stored).
PACK_CONSTANT = 0xFFFFFFFF


def pack_header_bytes(n_shard_tiles):
    return (n_shard_tiles * 8 + 1023) & ~1023


def packed_shard_bytes(W, H, world, tiles):
    st = shard_tiles(world, tiling(W, H)[2])
    return pack_header_bytes(st) + min(int(tiles), st) * 1024


def pack_packed(shard, rank, world, W, H, capacity_bytes):
    """This rank's shard bytes (pack_shard) -> (packed bytes [capacity_bytes], tiles stored, overflow flag).  Slots are
    handed out in tile order here (the kernel hands them out by an atomic counter: any order, the header says where)."""
    n_tiles = tiling(W, H)[2]
    st = shard_tiles(world, n_tiles)
    nl = local_tiles(rank, world, n_tiles)
    header_b = pack_header_bytes(st)
    max_slots = min((capacity_bytes - header_b) // 1024, st)
    out = np.zeros(capacity_bytes, np.uint8)
    header = out[: st * 8].view(np.uint32).reshape(st, 2)
    tiles = np.asarray(shard, np.uint8).reshape(st, 256, 4).view(np.uint32).reshape(st, 256)
    used, overflow = 0, 0
    for t in range(nl):
        px = tiles[t]
        if (px == px[0]).all():
            header[t] = (PACK_CONSTANT, px[0])
        elif used < max_slots:
            header[t] = (used, 0)
            out[header_b + used * 1024: header_b + (used + 1) * 1024] = px.view(np.uint8)
            used += 1
        else:
            header[t] = (PACK_CONSTANT, 0)
            overflow = 1
    return out, used, overflow


def assemble_packed(gathered, stride_bytes, W, H, world):
    """world packed shards, stride_bytes apart in rank order -> raster [H, W, 4] uint8 (volym_assemble_packed)."""
    n_tiles = tiling(W, H)[2]
    st = shard_tiles(world, n_tiles)
    header_b = pack_header_bytes(st)
    g = np.asarray(gathered, np.uint8).reshape(-1)
    idx = _tile_pixel_index(W, H)
    frame = np.zeros((H * W,), np.uint32)
    for k in range(n_tiles):
        r, local = k % world, k // world
        base = g[r * stride_bytes: (r + 1) * stride_bytes]
        slot, value = base[: st * 8].view(np.uint32).reshape(st, 2)[local]
        if slot == PACK_CONSTANT:
            px = np.full(256, value, np.uint32)
        else:
            px = base[header_b + int(slot) * 1024: header_b + (int(slot) + 1) * 1024].view(np.uint32)
        ok = idx[k] >= 0
        frame[idx[k][ok]] = px[ok]
    return frame.view(np.uint8).reshape(H, W, 4)
