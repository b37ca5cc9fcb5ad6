// Multi-GPU exchange from C: the sort-first tile gather and the sort-last depth composite over RCCL, so that a host
// without Python (the Rust engine) has them behind the same C ABI.  No counterpart in the reference,
// which is single-device.
//
// RCCL is loaded at rnd_comm_init (dlopen "librccl.so.1": the copy already in the process if there is one), so the
// library itself has no link-time dependency on it.  xGMI is point to point: both exchanges are groups of
// ncclSend / ncclRecv in which every peer uses its own link (no ring, no tree):
//   gather     every rank sends its tile-major RGBA8 buffer to rank 0, which scatters all of them into the frame;
//   composite  the packed (depth bits << 32 | rgba8) frame is cut into `world` strips, strip s of every rank goes to
//              rank s (all-to-all), rank s takes the element-wise minimum, the finished strips go to rank 0.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>

#include "visual_server.hpp"

namespace rnd {

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.lib) break;
        }
        if (!x.lib) return x;
        auto sym = [&](const char* n) { return dlsym(x.lib, n); };
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
        x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
        x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
        return x;
    }();
    if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd)
        throw Error(RND_ERR_UNSUPPORTED, "RCCL (librccl.so.1) could not be loaded");
    return r;
}

void check_nccl(ncclResult_t e, const char* what) {
    if (e == ncclSuccess) return;
    const char* text = rccl().GetErrorString ? rccl().GetErrorString(e) : "error";
    throw Error(RND_ERR_DEVICE, std::string(what) + ": " + text);
}
#define NCCL_CHECK(x) check_nccl((x), #x)
}  // namespace

static_assert(RND_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rnd.h carries the RCCL unique id by value");

void comm_make_id(uint8_t* id) {
    ncclUniqueId u;
    NCCL_CHECK(rccl().GetUniqueId(&u));
    std::memcpy(id, u.internal, RND_COMM_ID_BYTES);
}

void VisualServer::comm_init(const uint8_t* id, uint32_t rank, uint32_t world) {
    if (!id || world == 0 || rank >= world) throw Error(RND_ERR_INVALID, "bad communicator arguments");
    use_device();
    comm_destroy();
    ncclUniqueId u;
    std::memcpy(u.internal, id, RND_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    NCCL_CHECK(rccl().CommInitRank(&c, (int)world, u, (int)rank));
    comm_ = c; comm_rank_ = rank; comm_world_ = world;
}

void VisualServer::comm_destroy() {
    if (!comm_) return;
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    (void)rccl().CommDestroy(static_cast<ncclComm_t>(comm_));
    comm_ = nullptr; comm_world_ = 0; comm_rank_ = 0;
    d_comm_a_.release(); d_comm_b_.release();
}

// Sort-first: all ranks call it after rnd_render on the partition (rank, world) of the communicator.
void VisualServer::gather_tiles() {
    if (!comm_) throw Error(RND_ERR_INVALID, "no communicator: call rnd_comm_init first");
    if (world_ != comm_world_ || rank_ != comm_rank_) throw Error(RND_ERR_INVALID, "the tile partition does not match the communicator's rank and world");
    if (comm_world_ == 1) return;           // the frame is already whole and row-major
    if (!last_tiles_out_) throw Error(RND_ERR_NO_FRAME, "no frame rendered on the partition yet");
    use_device();
    finish();            // a truncated frame is re-run here: its tiles must not leave as they are
    Rccl& r = rccl();
    ncclComm_t c = static_cast<ncclComm_t>(comm_);
    // every rank's share padded to the largest (the first slot's, i.e. the last rank's: stripe_owner)
    uint64_t rows0 = 0;
    for (uint32_t first = 0; first < tiles_y_; first += kOwnerGroup * comm_world_) rows0 += std::min(kOwnerGroup, tiles_y_ - first);
    const size_t stride = (size_t)rows0 * tiles_x_ * kTilePixels * 4;
    const size_t mine = (size_t)n_owned_ * kTilePixels * 4;
    if (comm_rank_ == 0) {
        d_comm_a_.ensure((stride * comm_world_ + 7) / 8);
        uint8_t* all = reinterpret_cast<uint8_t*>(d_comm_a_.ptr);
        HIP_CHECK_COMM(hipMemcpyAsync(all, last_tiles_out_, mine, hipMemcpyDeviceToDevice, stream_));
        NCCL_CHECK(r.GroupStart());
        for (uint32_t peer = 1; peer < comm_world_; ++peer) {
            const size_t n = owned_tiles(peer, comm_world_).size() * (size_t)kTilePixels * 4;
            if (n) NCCL_CHECK(r.Recv(all + peer * stride, n, ncclUint8, (int)peer, c, stream_));
        }
        NCCL_CHECK(r.GroupEnd());
        detile_gathered(comm_world_, all, stride);
        assembled_frame_ = true;        // d_frame_ holds every rank's tiles: read-backs hand it out as it is
    } else if (mine) {
        NCCL_CHECK(r.GroupStart());
        NCCL_CHECK(r.Send(last_tiles_out_, mine, ncclUint8, 0, c, stream_));
        NCCL_CHECK(r.GroupEnd());
    }
}

// Sort-last: all ranks call it after rnd_render of their shard (packed output on: rnd_packed_device_ptr once).
void VisualServer::composite_packed() {
    if (!comm_) throw Error(RND_ERR_INVALID, "no communicator: call rnd_comm_init first");
    if (!want_packed_ || !last_packed_out_) throw Error(RND_ERR_NO_FRAME, "no packed frame: call rnd_packed_device_ptr once before rendering");
    use_device();
    finish();            // (as gather_tiles: a truncated shard is re-run before it is composited)
    Rccl& r = rccl();
    ncclComm_t c = static_cast<ncclComm_t>(comm_);
    const uint32_t W = comm_world_, me = comm_rank_;
    const uint64_t P = (uint64_t)width_ * height_;
    const uint64_t strip = ((P + W - 1) / W + 1) & ~1ull;            // words per strip, even (16-byte loads)
    auto len = [&](uint32_t s) { const uint64_t lo = (uint64_t)s * strip; return lo >= P ? 0ull : std::min(strip, P - lo); };
    const uint64_t* packed = last_packed_out_;
    // a: [W, strip] the strips of rank `me` as every rank rendered them, then (rank 0) b: [W * strip] the finished strips
    d_comm_a_.ensure((size_t)W * strip);
    d_comm_b_.ensure((size_t)(me == 0 ? W : 1) * strip);
    uint64_t* recv = d_comm_a_.ptr;
    uint64_t* done = d_comm_b_.ptr;
    const uint64_t my_len = len(me);
    if (my_len) HIP_CHECK_COMM(hipMemcpyAsync(recv + (size_t)me * strip, packed + (size_t)me * strip, my_len * 8, hipMemcpyDeviceToDevice, stream_));
    NCCL_CHECK(r.GroupStart());
    for (uint32_t peer = 0; peer < W; ++peer) {
        if (peer == me) continue;
        if (len(peer)) NCCL_CHECK(r.Send(packed + (size_t)peer * strip, len(peer), ncclUint64, (int)peer, c, stream_));
        if (my_len) NCCL_CHECK(r.Recv(recv + (size_t)peer * strip, my_len, ncclUint64, (int)peer, c, stream_));
    }
    NCCL_CHECK(r.GroupEnd());
    uint64_t* mine_out = done;      // (rank 0: its own strip is strip 0 of the finished frame)
    launch_min_packed(recv, W, strip, mine_out, my_len, stream_);
    NCCL_CHECK(r.GroupStart());
    if (me == 0) {
        for (uint32_t peer = 1; peer < W; ++peer)
            if (len(peer)) NCCL_CHECK(r.Recv(done + (size_t)peer * strip, len(peer), ncclUint64, (int)peer, c, stream_));
    } else if (my_len) {
        NCCL_CHECK(r.Send(mine_out, my_len, ncclUint64, 0, c, stream_));
    }
    NCCL_CHECK(r.GroupEnd());
    if (me == 0) {
        FrameConst fc = fc_;
        fc.width = width_; fc.height = height_;
        launch_unpack(done, fc, d_frame_.ptr, nullptr, stream_);
        last_frame_out_ = d_frame_.ptr;
        have_frame_ = true;
        assembled_frame_ = true;
    }
    HIP_CHECK_COMM(hipGetLastError());
}

}  // namespace rnd
