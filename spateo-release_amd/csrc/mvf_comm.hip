// The EM step's exchange over RCCL (xGMI): an opaque communicator + the in-place float64 all-reduce of the step's
// sufficient statistics.  No kernels of our own here: the reduction IS the library collective (ring / tree over the
// point-to-point xGMI links); what this file owns is the boundary - plain C, caller-owned device buffer, caller's stream.
#include "mvf_common.h"

#include <rccl/rccl.h>

#include <cstring>
#include <new>

struct mvf_comm {
    ncclComm_t comm;
    int nranks, rank, device;
};

namespace {

int nccl_fail(const char* what, ncclResult_t r) { return mvf::set_error("%s failed: %s", what, ncclGetErrorString(r)); }

}  // namespace

extern "C" int mvf_comm_unique_id(void* id_out) {
    MVF_REQUIRE(id_out != nullptr, "mvf_comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == MVF_COMM_ID_BYTES, "mvf.h: MVF_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
    std::memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int mvf_comm_create(mvf_comm** comm_out, int nranks, int rank, const void* id) {
    MVF_REQUIRE(comm_out != nullptr && id != nullptr, "mvf_comm_create: null pointer");
    MVF_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "mvf_comm_create: rank %d outside [0, %d)", rank, nranks);
    *comm_out = nullptr;
    int dev = 0;
    MVF_CHECK_HIP(hipGetDevice(&dev));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    ncclResult_t r = ncclCommInitRank(&c, nranks, uid, rank);  // blocks until all nranks ranks have called it
    if (r != ncclSuccess) return nccl_fail("ncclCommInitRank", r);
    mvf_comm* h = new (std::nothrow) mvf_comm{c, nranks, rank, dev};
    if (!h) {
        ncclCommDestroy(c);
        return mvf::set_error("mvf_comm_create: out of host memory");
    }
    *comm_out = h;
    return 0;
}

extern "C" int mvf_comm_destroy(mvf_comm* comm) {
    if (!comm) return 0;
    ncclResult_t r = ncclCommDestroy(comm->comm);
    delete comm;
    if (r != ncclSuccess) return nccl_fail("ncclCommDestroy", r);
    return 0;
}

extern "C" int mvf_comm_info(const mvf_comm* comm, int* nranks, int* rank, int* device) {
    MVF_REQUIRE(comm != nullptr, "mvf_comm_info: null communicator");
    // what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), not the arguments
    // it was created with: the figure a scaling record can cite as "N ranks took part"
    int n = -1, r = -1, d = -1;
    ncclResult_t e = ncclCommCount(comm->comm, &n);
    if (e != ncclSuccess) return nccl_fail("ncclCommCount", e);
    e = ncclCommUserRank(comm->comm, &r);
    if (e != ncclSuccess) return nccl_fail("ncclCommUserRank", e);
    e = ncclCommCuDevice(comm->comm, &d);
    if (e != ncclSuccess) return nccl_fail("ncclCommCuDevice", e);
    MVF_REQUIRE(n == comm->nranks && r == comm->rank, "mvf_comm_info: RCCL reports rank %d of %d, the handle was created as %d of %d",
                r, n, comm->rank, comm->nranks);
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (device) *device = d;
    return 0;
}

extern "C" int mvf_allreduce_stats(mvf_comm* comm, double* buf, int64_t count, int op, void* stream) {
    MVF_REQUIRE(comm != nullptr, "mvf_allreduce_stats: null communicator");
    MVF_REQUIRE(count >= 0, "mvf_allreduce_stats: negative count");
    MVF_REQUIRE(op == MVF_RED_SUM || op == MVF_RED_MIN, "mvf_allreduce_stats: op must be MVF_RED_SUM or MVF_RED_MIN");
    if (count == 0) return 0;
    MVF_REQUIRE(buf != nullptr, "mvf_allreduce_stats: null buffer");
    int dev = 0;
    MVF_CHECK_HIP(hipGetDevice(&dev));
    MVF_REQUIRE(dev == comm->device, "mvf_allreduce_stats: current device %d is not the communicator's device %d", dev,
                comm->device);
    ncclResult_t r = ncclAllReduce(buf, buf, (size_t)count, ncclDouble, op == MVF_RED_SUM ? ncclSum : ncclMin, comm->comm,
                                   (hipStream_t)stream);
    if (r != ncclSuccess) return nccl_fail("ncclAllReduce", r);
    return 0;
}
