// Deterministic synthetic stand-ins for the reference's missing .raw assets, identical byte for byte to
// volym_amd/synth.py (integer-only arithmetic; tests pin both to the same SHA-256).
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/volym_host.h"

namespace {

inline uint32_t lowbias32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

inline uint32_t voxel_hash(uint32_t nx, uint32_t ny, uint32_t x, uint32_t y, uint32_t z, uint32_t seed)
{
    return lowbias32(x + nx * (y + ny * z) + seed);
}

struct Sphere { int64_t cx, cy, cz, r; };

}  // namespace

extern "C" {

int volym_synth_bonsai(uint32_t n, uint32_t seed, uint8_t* density, uint8_t* labels)
{
    if (!density || n == 0 || n > 2048) return VOLYM_E_INVALID;
    const int64_t N = n;
    Sphere sp[12];
    for (int i = 0; i < 12; ++i) {
        const uint32_t h0 = lowbias32(seed * 31u + 4u * i + 0u), h1 = lowbias32(seed * 31u + 4u * i + 1u);
        const uint32_t h2 = lowbias32(seed * 31u + 4u * i + 2u), h3 = lowbias32(seed * 31u + 4u * i + 3u);
        sp[i].cx = N / 4 + static_cast<int64_t>(h0 % 1024u) * (N / 2) / 1024;
        sp[i].cy = (52 * N) / 100 + static_cast<int64_t>(h1 % 1024u) * ((32 * N) / 100) / 1024;
        sp[i].cz = N / 4 + static_cast<int64_t>(h2 % 1024u) * (N / 2) / 1024;
        sp[i].r = (10 * N) / 100 + static_cast<int64_t>(h3 % 1024u) * ((9 * N) / 100) / 1024;
        if (sp[i].r < 1) sp[i].r = 1;
    }
    const int64_t half = N / 2, pot_y0 = (5 * N) / 100, pot_y1 = (18 * N) / 100, pot_h = (22 * N) / 100;
    const int64_t trunk_y1 = (55 * N) / 100, trunk_r = (45 * N) / 1000, trunk_r2 = trunk_r * trunk_r;
    for (int64_t z = 0; z < N; ++z)
        for (int64_t y = 0; y < N; ++y)
            for (int64_t x = 0; x < N; ++x) {
                const uint32_t h = voxel_hash(n, n, static_cast<uint32_t>(x), static_cast<uint32_t>(y), static_cast<uint32_t>(z), seed);
                int64_t s = h & 7u;
                uint8_t l = 0;
                for (int i = 0; i < 12; ++i) {
                    const int64_t dx = x - sp[i].cx, dy = y - sp[i].cy, dz = z - sp[i].cz, r2 = sp[i].r * sp[i].r;
                    const int64_t d2 = dx * dx + dy * dy + dz * dz;
                    if (d2 < r2) {
                        const int64_t noise = static_cast<int64_t>((h >> 12) & 15u) - 8;
                        int64_t q = (70 * d2) / r2;      // d2 >= 0: floor == truncation
                        const int64_t val = 160 - q + noise;
                        if (val > s) { s = val; l = 2; }
                    }
                }
                const int64_t dzt = z - half;
                if ((x - half) * (x - half) + dzt * dzt < trunk_r2 && y >= pot_y1 && y < trunk_y1) { s = 192 + ((h >> 8) & 15u); l = 3; }
                if (y >= pot_y0 && y < pot_y1 && std::abs(x - half) < pot_h && std::abs(dzt) < pot_h) { s = 230; l = 4; }
                const size_t o = static_cast<size_t>(x) + static_cast<size_t>(N) * (y + static_cast<size_t>(N) * z);
                density[o] = static_cast<uint8_t>(s);
                if (labels) labels[o] = l;
            }
    return VOLYM_OK;
}

int volym_synth_teapot(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t seed, uint8_t* density, uint8_t* labels)
{
    if (!density || !labels || nx == 0 || ny == 0 || nz == 0) return VOLYM_E_INVALID;
    const int64_t NX = nx, NY = ny, NZ = nz;
    const int64_t cx = NX / 2, cy = (47 * NY) / 100, cz = NZ / 2;
    const int64_t ro = (27 * NX) / 100, ri = (23 * NX) / 100, g0 = (10 * NY) / 100, g1 = (16 * NY) / 100;
    const int64_t lx = (13 * NX) / 100, ly = (8 * NY) / 100, lz = (11 * NX) / 100, lcy = (42 * NY) / 100;
    for (int64_t z = 0; z < NZ; ++z)
        for (int64_t y = 0; y < NY; ++y)
            for (int64_t x = 0; x < NX; ++x) {
                const uint32_t h = voxel_hash(nx, ny, static_cast<uint32_t>(x), static_cast<uint32_t>(y), static_cast<uint32_t>(z), seed + 1u);
                int64_t s = h & 7u;
                uint8_t l = 0;
                const int64_t dz = z - cz;
                const int64_t d2 = (x - cx) * (x - cx) + (y - cy) * (y - cy) + dz * dz;
                if (d2 < ro * ro && d2 >= ri * ri && y >= g1 && y < cy + (15 * NY) / 100) { s = 104 + ((h >> 8) & 15u); l = 3; }
                const int64_t e = (x - cx) * (x - cx) * (ly * ly * lz * lz) + (y - lcy) * (y - lcy) * (lx * lx * lz * lz) + dz * dz * (lx * lx * ly * ly);
                if (e < lx * lx * ly * ly * lz * lz) { s = 192 + ((h >> 16) & 15u); l = 2; }
                if (y >= g0 && y < g1) { s = 66 + ((h >> 20) & 7u); l = 4; }
                const size_t o = static_cast<size_t>(x) + static_cast<size_t>(NX) * (y + static_cast<size_t>(NY) * z);
                density[o] = static_cast<uint8_t>(s);
                labels[o] = l;
            }
    return VOLYM_OK;
}

}  // extern "C"
