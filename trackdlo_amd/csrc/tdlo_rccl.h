// tdlo_rccl.h -- RCCL's C API, bound at run time.
//
// The N-split driver (tdlo_split_run) calls RCCL directly -- no torch, no Python -- but libtrackdlo_hip.so does not carry a
// DT_NEEDED on librccl: single-GPU users need no RCCL at all, and a process that also holds PyTorch must use the ONE RCCL
// PyTorch loaded (its wheel ships its own librccl.so next to its own HIP runtime; two RCCLs in one process means two sets of
// communicator state).  The library is therefore resolved on first use: an already mapped librccl wins, then $TDLO_RCCL_LIB
// or the path given to tdlo_rccl_load, then the system's librccl.so.1.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <string>

namespace tdlo {

struct RcclApi {
    // the subset of rccl.h this library uses (types reduced to what crosses the call: ncclComm_t is an opaque pointer,
    // ncclUniqueId is 128 bytes passed BY VALUE to ncclCommInitRank, enums are ints)
    struct UniqueId { char internal[128]; };
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **comm, int nranks, UniqueId id, int rank) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*CommCount)(void *comm, int *count) = nullptr;
    int (*CommUserRank)(void *comm, int *rank) = nullptr;
    int (*AllReduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t s) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    void *handle = nullptr;
    std::string path;
};
constexpr int kNcclFloat64 = 8, kNcclSum = 0, kNcclMin = 3;      // ncclDataType_t / ncclRedOp_t values of rccl.h (checked in tdlo_rccl.cpp)

// nullptr when no RCCL can be loaded; *why receives the reason
const RcclApi *rccl_api(const char *path_hint, std::string *why);

}  // namespace tdlo
