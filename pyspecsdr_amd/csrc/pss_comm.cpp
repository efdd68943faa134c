// pss_comm.cpp — the path's exchange steps behind the C ABI (include/pss.h "Multi-GPU"): one process per GPU, RCCL over xGMI.
//
// The hot path shards by contiguous blocks of frames / scanner slices with NO data-path collective (pyspecsdr.py:2514-2590: every slice
// of a sweep, every read buffer of the loop is processed on its own).  What crosses ranks is
//   (1) the gather of a rank's packed results to one rank (or to all): pss_gather_packed — the Python host's shard.gather_packed;
//   (2) for the display accumulators, the row extremes of the frames just before a rank's block: pss_halo_from_left — shard.halo_from_left.
// A Python host does both over torch.distributed (pyspecsdr_amd/shard.py, backend "nccl" = RCCL); these entry points give a host
// WITHOUT torch the same two steps on the context's stream.  librccl is opened at the first call (dlopen: the library this file is
// linked into has no load-time dependency on it, and a process that already holds an RCCL — torch's — shares that copy); the types
// come from <rccl/rccl.h>, the functions from dlsym.
// Rendezvous is the host's business, as with MPI / NCCL everywhere: rank 0 calls pss_comm_id, hands the 128 bytes to the other
// ranks by whatever it has (a file, a socket, its own launcher), every rank calls pss_comm_init.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "pss_ctx.h"

static_assert(PSS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "pss.h's rendezvous id is RCCL's ncclUniqueId");

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

// the process's RCCL: $PSS_RCCL_LIB if the host names one (an explicit choice wins — also over a copy the process has loaded already: the tests
// put a transport double there, tests/rccl_double/), else the copy already loaded (RTLD_NOLOAD matches torch's bundled librccl.so by its
// SONAME), the loader's search path, /opt/rocm/lib
Rccl *rccl()
{
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (r.lib) return &r;
    const char *env = std::getenv("PSS_RCCL_LIB");
    void *h = (env && *env) ? dlopen(env, RTLD_NOW | RTLD_LOCAL) : nullptr;
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        const char *e = dlerror();
        r.err = std::string("librccl.so.1 not found (set PSS_RCCL_LIB): ") + (e ? e : "dlopen failed");
        return &r;
    }
    bool ok = true;
    auto sym = [&](auto &fp, const char *name) {
        fp = reinterpret_cast<std::remove_reference_t<decltype(fp)>>(dlsym(h, name));
        if (!fp) { ok = false; r.err = std::string("librccl: symbol ") + name + " missing"; }
    };
    sym(r.GetUniqueId, "ncclGetUniqueId");
    sym(r.CommInitRank, "ncclCommInitRank");
    sym(r.CommDestroy, "ncclCommDestroy");
    sym(r.AllGather, "ncclAllGather");
    sym(r.Send, "ncclSend");
    sym(r.Recv, "ncclRecv");
    sym(r.GroupStart, "ncclGroupStart");
    sym(r.GroupEnd, "ncclGroupEnd");
    sym(r.GetErrorString, "ncclGetErrorString");
    if (ok) r.lib = h;
    return &r;
}

int nccl_check(pss_ctx *ctx, Rccl *r, ncclResult_t e, const char *what)
{
    if (e == ncclSuccess) return PSS_OK;
    return pss_fail(ctx, PSS_E_COMM, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "RCCL error"));
}
#define PSS_NCCL(ctx, r, call)                                  \
    do {                                                        \
        int _q = nccl_check((ctx), (r), (call), #call);         \
        if (_q) return _q;                                      \
    } while (0)

}  // namespace

// contiguous blocks whose sizes differ by at most one — shard.shard_range
extern "C" int pss_shard_range(long n_items, int rank, int n_ranks, long *start, long *count)
{
    if (n_items < 0 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return PSS_E_ARG;
    const long base = n_items / n_ranks, rem = n_items % n_ranks;
    if (start) *start = rank * base + (rank < rem ? rank : rem);
    if (count) *count = base + (rank < rem ? 1 : 0);
    return PSS_OK;
}

extern "C" int pss_comm_id(void *id)
{
    if (!id) return PSS_E_ARG;
    Rccl *r = rccl();
    if (!r->lib) return PSS_E_COMM;
    ncclUniqueId u;
    if (r->GetUniqueId(&u) != ncclSuccess) return PSS_E_COMM;
    std::memcpy(id, u.internal, PSS_COMM_ID_BYTES);
    return PSS_OK;
}

extern "C" int pss_comm_init(pss_ctx *ctx, const void *id, int rank, int n_ranks)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return pss_fail(ctx, PSS_E_ARG, "pss_comm_init: rank outside [0, n_ranks)");
    if (ctx->comm) return pss_fail(ctx, PSS_E_ARG, "pss_comm_init: the context already has a communicator (pss_comm_free first)");
    if (n_ranks > 1 && !id) return pss_fail(ctx, PSS_E_ARG, "pss_comm_init: rendezvous id missing");
    if (n_ranks == 1 && !id) {      // a lone rank needs no RCCL at all: the exchange steps degenerate to copies
        ctx->comm_rank = 0;
        ctx->comm_n = 1;
        return PSS_OK;
    }
    Rccl *r = rccl();
    if (!r->lib) return pss_fail(ctx, PSS_E_COMM, r->err);
    ncclUniqueId u;
    std::memcpy(u.internal, id, PSS_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    PSS_NCCL(ctx, r, r->CommInitRank(&c, n_ranks, u, rank));
    ctx->comm = c;
    ctx->comm_rank = rank;
    ctx->comm_n = n_ranks;
    return PSS_OK;
}

extern "C" int pss_comm_free(pss_ctx *ctx)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    int q = PSS_OK;
    if (ctx->comm) {
        Rccl *r = rccl();
        hipStreamSynchronize(ctx->stream);
        if (r->lib) q = nccl_check(ctx, r, r->CommDestroy(static_cast<ncclComm_t>(ctx->comm)), "ncclCommDestroy");
        ctx->comm = nullptr;
    }
    ctx->comm_rank = 0;
    ctx->comm_n = 1;
    return q;
}

extern "C" int pss_comm_size(pss_ctx *ctx, int *rank, int *n_ranks)
{
    if (!ctx) return PSS_E_ARG;
    if (rank) *rank = ctx->comm_rank;
    if (n_ranks) *n_ranks = ctx->comm_n;
    return PSS_OK;
}

// ONE collective for everything a rank produced in a sharded pass.  dst < 0: all-gather (ncclAllGather).  dst >= 0: a gather to that rank
// as grouped point-to-point messages — on the xGMI mesh every peer's message takes that peer's own link to the root, nothing is staged
// through a ring.
extern "C" int pss_gather_packed(pss_ctx *ctx, const void *d_local, size_t bytes, void *d_all, int dst)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    const int n = ctx->comm_n, me = ctx->comm_rank;
    if (dst >= n) return pss_fail(ctx, PSS_E_ARG, "pss_gather_packed: dst outside the communicator");
    const bool recv = dst < 0 || dst == me;
    if (bytes > 0 && (!d_local || (recv && !d_all))) return pss_fail(ctx, PSS_E_ARG, "pss_gather_packed: null buffer");
    if (bytes == 0) return PSS_OK;
    hipStream_t st = PSS_STREAM(ctx);
    char *mine = recv ? static_cast<char *>(d_all) + (size_t)me * bytes : nullptr;
    if (!ctx->comm) {       // a lone rank without RCCL: the gather is a copy (a one-rank communicator takes the RCCL calls below like any other)
        if (n != 1) return pss_fail(ctx, PSS_E_COMM, "pss_gather_packed: no communicator");
        if (mine != d_local) PSS_HIP(ctx, hipMemcpyAsync(mine, d_local, bytes, hipMemcpyDeviceToDevice, st));
        return PSS_OK;
    }
    Rccl *r = rccl();
    ncclComm_t c = static_cast<ncclComm_t>(ctx->comm);
    if (dst < 0) {
        PSS_NCCL(ctx, r, r->AllGather(d_local, d_all, bytes, ncclUint8, c, st));
        return PSS_OK;
    }
    if (me == dst && mine != d_local) PSS_HIP(ctx, hipMemcpyAsync(mine, d_local, bytes, hipMemcpyDeviceToDevice, st));
    PSS_NCCL(ctx, r, r->GroupStart());
    ncclResult_t e = ncclSuccess;
    if (me == dst) {
        for (int p = 0; p < n && e == ncclSuccess; p++)
            if (p != me) e = r->Recv(static_cast<char *>(d_all) + (size_t)p * bytes, bytes, ncclUint8, p, c, st);
    } else {
        e = r->Send(d_local, bytes, ncclUint8, dst, c, st);
    }
    const ncclResult_t e2 = r->GroupEnd();
    PSS_NCCL(ctx, r, e);
    PSS_NCCL(ctx, r, e2);
    return PSS_OK;
}

// The rows that precede this rank's block in global order, as far back as `halo` rows: rank r wants the global rows
// [max(0, start_r - halo), start_r).  counts[] (every rank's block size, the same array on every rank) tells each rank which of its
// rows which rank to its right needs and which ranks to its left hold its own halo (a block shorter than `halo` makes the halo span
// several neighbours); all messages of a rank are posted in one group, so no rank waits for a neighbour's receive before it sends.
extern "C" int pss_halo_from_left(pss_ctx *ctx, const void *d_rows, const long *counts, size_t row_bytes, long halo, void *d_halo,
                                  long *n_halo)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_halo) *n_halo = 0;
    const int n = ctx->comm_n, me = ctx->comm_rank;
    if (halo < 0) return pss_fail(ctx, PSS_E_ARG, "pss_halo_from_left: halo < 0");
    if (n == 1 || halo == 0 || row_bytes == 0) return PSS_OK;
    if (!ctx->comm) return pss_fail(ctx, PSS_E_COMM, "pss_halo_from_left: no communicator");
    if (!counts) return pss_fail(ctx, PSS_E_ARG, "pss_halo_from_left: block sizes missing");
    std::vector<long> start(n + 1, 0);
    for (int p = 0; p < n; p++) {
        if (counts[p] < 0) return pss_fail(ctx, PSS_E_ARG, "pss_halo_from_left: negative block size");
        start[p + 1] = start[p] + counts[p];
    }
    auto need_lo = [&](int p) { return start[p] - halo > 0 ? start[p] - halo : 0L; };
    const long lo = need_lo(me), hi = start[me];
    if (hi > lo && !d_halo) return pss_fail(ctx, PSS_E_ARG, "pss_halo_from_left: null halo buffer");
    if (counts[me] > 0 && !d_rows) return pss_fail(ctx, PSS_E_ARG, "pss_halo_from_left: null rows");
    Rccl *r = rccl();
    ncclComm_t c = static_cast<ncclComm_t>(ctx->comm);
    hipStream_t st = PSS_STREAM(ctx);
    PSS_NCCL(ctx, r, r->GroupStart());
    ncclResult_t e = ncclSuccess;
    for (int s = 0; s < me && e == ncclSuccess; s++) {            // what I receive, left to right
        const long a = lo > start[s] ? lo : start[s], b = hi < start[s + 1] ? hi : start[s + 1];
        if (b > a) e = r->Recv(static_cast<char *>(d_halo) + (size_t)(a - lo) * row_bytes, (size_t)(b - a) * row_bytes, ncclUint8, s, c, st);
    }
    for (int p = me + 1; p < n && e == ncclSuccess; p++) {        // what the ranks to my right need from me
        long a = need_lo(p), b = start[p];
        a = a > start[me] ? a : start[me];
        b = b < start[me + 1] ? b : start[me + 1];
        if (b > a)
            e = r->Send(static_cast<const char *>(d_rows) + (size_t)(a - start[me]) * row_bytes, (size_t)(b - a) * row_bytes, ncclUint8, p, c, st);
    }
    const ncclResult_t e2 = r->GroupEnd();
    PSS_NCCL(ctx, r, e);
    PSS_NCCL(ctx, r, e2);
    if (n_halo) *n_halo = hi - lo;
    return PSS_OK;
}
