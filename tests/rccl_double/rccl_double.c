/* rccl_double.c — TEST INFRASTRUCTURE: a transport double for the nine RCCL entry points libpss.so's exchange steps use (pss_comm.cpp), so that
 * pss_gather_packed's grouped send / receive and pss_halo_from_left's multi-neighbour case can run with REAL peer processes on a box that has
 * one GPU (RCCL itself refuses two ranks on one device).  The product never loads this: pss_comm.cpp opens librccl, and only a process whose
 * environment points PSS_RCCL_LIB at this file — and that has no RCCL loaded already — gets the double (tests/test_gpu_parity.py builds it
 * into a temporary directory).  What it keeps of RCCL's semantics: the call signatures (<rccl/rccl.h>), group semantics (operations posted
 * between ncclGroupStart / ncclGroupEnd are issued together, sends never wait for their receives), FIFO order per (source, destination)
 * pair, byte counts checked at the receiver.  What it replaces: the transport — a message is a file in $PSS_RCCL_DOUBLE_DIR/<id>/ (written
 * under a temporary name and renamed: atomic publish), moved with synchronous hipMemcpy on both sides.
 *   gcc -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/rccl_double/rccl_double.c -L/opt/rocm/lib -lamdhip64 -o librccl_double.so */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define MAX_RANKS 64
#define MAX_OPS 1024

struct ncclComm {
    int rank, n;
    char dir[768];
    unsigned long seq_send[MAX_RANKS], seq_recv[MAX_RANKS];
};

struct op { int send; void *buf; size_t bytes; int peer; struct ncclComm *c; hipStream_t st; };
static __thread struct op g_ops[MAX_OPS];
static __thread int g_n_ops = 0, g_depth = 0;

static ncclResult_t do_send(struct ncclComm *c, const void *buf, size_t bytes, int peer, hipStream_t st)
{
    if (peer < 0 || peer >= c->n || peer == c->rank) return ncclInvalidArgument;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    void *h = malloc(bytes ? bytes : 1);
    if (!h) return ncclSystemError;
    if (hipMemcpy(h, buf, bytes, hipMemcpyDeviceToHost) != hipSuccess) { free(h); return ncclUnhandledCudaError; }
    char tmp[1100], fin[1024];
    snprintf(fin, sizeof fin, "%s/msg_%d_%d_%lu", c->dir, c->rank, peer, c->seq_send[peer]);
    snprintf(tmp, sizeof tmp, "%s.tmp", fin);
    FILE *f = fopen(tmp, "wb");
    const int bad = !f || fwrite(h, 1, bytes, f) != bytes || fclose(f) || rename(tmp, fin);
    free(h);
    if (bad) return ncclSystemError;
    c->seq_send[peer]++;
    return ncclSuccess;
}

static ncclResult_t do_recv(struct ncclComm *c, void *buf, size_t bytes, int peer, hipStream_t st)
{
    if (peer < 0 || peer >= c->n || peer == c->rank) return ncclInvalidArgument;
    char fin[1024];
    snprintf(fin, sizeof fin, "%s/msg_%d_%d_%lu", c->dir, peer, c->rank, c->seq_recv[peer]);
    FILE *f = NULL;
    for (int tries = 0; tries < 120000 && !(f = fopen(fin, "rb")); tries++) usleep(1000);
    if (!f) return ncclSystemError;                       /* the peer never sent (two minutes) */
    fseek(f, 0, SEEK_END);
    const long have = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (have < 0 || (size_t)have != bytes) { fclose(f); return ncclInvalidArgument; }     /* the two sides disagree about the message size */
    void *h = malloc(bytes ? bytes : 1);
    const int bad = !h || fread(h, 1, bytes, f) != bytes;
    fclose(f);
    unlink(fin);
    if (bad) { free(h); return ncclSystemError; }
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(buf, h, bytes, hipMemcpyHostToDevice);
    free(h);
    if (e != hipSuccess) return ncclUnhandledCudaError;
    c->seq_recv[peer]++;
    return ncclSuccess;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id->internal, 0, sizeof id->internal);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "pssdbl-%d-%ld-%ld", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    const char *base = getenv("PSS_RCCL_DOUBLE_DIR");
    if (!base || !*base) return ncclInvalidUsage;
    struct ncclComm *c = (struct ncclComm *)calloc(1, sizeof *c);
    if (!c) return ncclSystemError;
    c->rank = rank;
    c->n = nranks;
    char name[128];
    memcpy(name, id.internal, sizeof name - 1);
    name[sizeof name - 1] = 0;
    snprintf(c->dir, sizeof c->dir, "%s/%s", base, name);
    mkdir(c->dir, 0700);                                   /* (every rank tries; one wins) */
    char here[1024];
    snprintf(here, sizeof here, "%s/here_%d", c->dir, rank);
    FILE *f = fopen(here, "wb");
    if (!f) { free(c); return ncclSystemError; }
    fclose(f);
    for (int p = 0; p < nranks; p++) {                     /* the initialisation is collective: wait for every rank */
        snprintf(here, sizeof here, "%s/here_%d", c->dir, p);
        int tries = 0;
        while (access(here, F_OK) != 0 && tries++ < 120000) usleep(1000);
        if (access(here, F_OK) != 0) { free(c); return ncclSystemError; }
    }
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    free(comm);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void)
{
    g_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void)
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    ncclResult_t r = ncclSuccess;
    for (int i = 0; i < g_n_ops && r == ncclSuccess; i++)          /* every send of the group first: none of them waits for anybody */
        if (g_ops[i].send) r = do_send(g_ops[i].c, g_ops[i].buf, g_ops[i].bytes, g_ops[i].peer, g_ops[i].st);
    for (int i = 0; i < g_n_ops && r == ncclSuccess; i++)
        if (!g_ops[i].send) r = do_recv(g_ops[i].c, g_ops[i].buf, g_ops[i].bytes, g_ops[i].peer, g_ops[i].st);
    g_n_ops = 0;
    return r;
}

static size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

static ncclResult_t post(int send, void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st)
{
    const size_t tb = type_bytes(t);
    if (!comm || !tb) return ncclInvalidArgument;
    if (g_depth > 0) {
        if (g_n_ops >= MAX_OPS) return ncclInternalError;
        g_ops[g_n_ops++] = (struct op){send, buf, count * tb, peer, comm, st};
        return ncclSuccess;
    }
    return send ? do_send(comm, buf, count * tb, peer, st) : do_recv(comm, buf, count * tb, peer, st);
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(1, (void *)sendbuff, count, datatype, peer, comm, stream);
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(0, recvbuff, count, datatype, peer, comm, stream);
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    const size_t bytes = sendcount * type_bytes(datatype);
    if (!comm || !bytes) return ncclInvalidArgument;
    char *mine = (char *)recvbuff + (size_t)comm->rank * bytes;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (mine != (const char *)sendbuff && hipMemcpy(mine, sendbuff, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t r = ncclSuccess;
    for (int p = 0; p < comm->n && r == ncclSuccess; p++)
        if (p != comm->rank) r = do_send(comm, sendbuff, bytes, p, stream);
    for (int p = 0; p < comm->n && r == ncclSuccess; p++)
        if (p != comm->rank) r = do_recv(comm, (char *)recvbuff + (size_t)p * bytes, bytes, p, stream);
    return r;
}

const char *ncclGetErrorString(ncclResult_t result)
{
    switch (result) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "transport double: HIP error";
    case ncclSystemError: return "transport double: file / timeout error";
    case ncclInvalidArgument: return "transport double: invalid argument (peer, or the two sides disagree about a message's size)";
    case ncclInvalidUsage: return "transport double: invalid usage (PSS_RCCL_DOUBLE_DIR unset, or ncclGroupEnd without ncclGroupStart)";
    default: return "transport double: error";
    }
}
