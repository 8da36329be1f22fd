/*
 * volym_mgpu.h -- native multi-GPU frame loop of the ray-march path (libvolym_hip.so).
 *
 * The reference has no multi-GPU path; this is the sharding BASELINE.json's north_star asks for:
 * the framebuffer is dealt by interleaved 16x16 screen tiles over the GPUs of one node, the volume is replicated, and
 * every frame's shards are gathered onto the root GPU over RCCL/xGMI -- grouped direct sends to the root (ncclSend /
 * ncclRecv inside one ncclGroupStart/End: the root's inbound traffic uses all its xGMI links in parallel; a ring would push
 * every byte through every link) -- and assembled into the raster there.  Shards travel PACKED: only the tiles that are not
 * constant (volym_pack_shard).
 *
 * The whole frame loop is native: one call runs N frames {march, pack, send/recv, assemble} with the collective of frame i
 * overlapping the march of frames i+1.. (rotating buffers, a compute and a communication stream per device), and, when the
 * view is static, replays it from a captured HIP graph (one cycle of the rotating buffers per replay).  No Python, no
 * per-frame host logic beyond the enqueues.
 *
 * Two ways to get the devices:
 *   volym_mgpu_create       one process, N devices, ncclCommInitAll;
 *   volym_mgpu_create_rank  one process per device (the launch `python -m torch.distributed.run` gives bench.py):
 *                           ncclCommInitRank with the 128-byte id rank 0 got from volym_mgpu_unique_id and handed to the
 *                           others by any means (bench.py: torch.distributed).
 * VOLYM_MGPU_COPY replaces RCCL by device-to-device copies onto the root (hipMemcpyAsync; peer copies between devices of
 * one process).  With it the same device may be listed several times: "virtual ranks", which is how the protocol is tested
 * on a one-GPU box.
 *
 * RCCL is loaded on first use (dlopen librccl.so.1): the single-GPU library has no link-time dependency on it.
 * Conventions as in volym_hip.h: 0 or a negative VOLYM_E_* code, volym_mgpu_last_error for the text.
 */
#ifndef VOLYM_MGPU_H
#define VOLYM_MGPU_H

#include <stddef.h>
#include <stdint.h>

#include "volym_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { VOLYM_MGPU_RCCL = 0, VOLYM_MGPU_COPY = 1 };

typedef struct volym_mgpu volym_mgpu;

/* where the time of a frame goes: each stage timed alone with HIP events on an otherwise idle device (volym_mgpu_profile) */
typedef struct volym_mgpu_split {
    float march_ms, pack_ms, collective_ms, assemble_ms;   /* this process's slowest local rank / the root */
} volym_mgpu_split;

typedef struct volym_mgpu_timing {
    double wall_ms;              /* first enqueue .. every stream of every local device drained */
    double enqueue_us_per_frame; /* host time spent enqueueing, per frame and local device */
    uint32_t frames;
    uint32_t graph_replays;      /* > 0: the frames were replayed from a captured HIP graph */
    uint32_t msg_bytes;          /* packed shard per rank and frame */
    uint32_t overflowed;         /* a packed shard ran out of room (the frame is then wrong: re-run volym_mgpu_prepare) */
} volym_mgpu_timing;

int volym_mgpu_unique_id(uint8_t id[128]);
int volym_mgpu_create(volym_mgpu** out, uint32_t width, uint32_t height, int n_devices, const int* device_ids, int transport);
int volym_mgpu_create_rank(volym_mgpu** out, uint32_t width, uint32_t height, int device_id, int rank, int world,
                           const uint8_t id[128]);
void volym_mgpu_destroy(volym_mgpu* mg);
const char* volym_mgpu_last_error(const volym_mgpu* mg);

int volym_mgpu_world(const volym_mgpu* mg);
int volym_mgpu_local_count(const volym_mgpu* mg);
/* the single-GPU context of local device i (rank volym_mgpu_local_rank(i)), for options, stats and checks */
volym_ctx* volym_mgpu_context(volym_mgpu* mg, int i);
int volym_mgpu_local_rank(const volym_mgpu* mg, int i);

/* the volym_set_* / volym_update of volym_hip.h, applied to every local context (the volume is replicated) */
int volym_mgpu_set_volume(volym_mgpu* mg, const uint8_t* voxels, uint32_t nx, uint32_t ny, uint32_t nz, int filter);
int volym_mgpu_set_importances(volym_mgpu* mg, const uint8_t* importances, uint32_t nx, uint32_t ny, uint32_t nz);
int volym_mgpu_set_transfer_function(volym_mgpu* mg, const uint8_t* rgba8, uint32_t n);
int volym_mgpu_set_option(volym_mgpu* mg, int key, int value);
int volym_mgpu_update(volym_mgpu* mg, const volym_camera_uniforms* camera, const volym_parameter_uniforms* parameters);

/* Size the packed messages for the current view: one untimed frame per rank, the maximum number of stored tiles over all
 * ranks (+ `slack_percent` room for a moving view), buffers allocated.  Collective: every process calls it.  Blocks. */
int volym_mgpu_prepare(volym_mgpu* mg, uint32_t slack_percent);
/* Run `frames` frames of the current view.  use_graph != 0: replay a captured HIP graph when possible (static view; falls
 * back to plain enqueues when capture is refused).  Blocks until the last frame is assembled on the root. */
int volym_mgpu_run(volym_mgpu* mg, uint32_t frames, int use_graph, volym_mgpu_timing* timing);
/* per-stage times (untimed diagnostic pass, stages serialised) */
int volym_mgpu_profile(volym_mgpu* mg, uint32_t frames, volym_mgpu_split* split);
/* the assembled frame; valid in the process that holds the root rank (rank 0), VOLYM_E_STATE elsewhere */
int volym_mgpu_read_rgba8(volym_mgpu* mg, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif
