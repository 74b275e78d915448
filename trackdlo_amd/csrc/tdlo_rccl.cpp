// tdlo_rccl.cpp -- run-time binding of RCCL (see tdlo_rccl.h).
#include "tdlo_rccl.h"

#include <dlfcn.h>
#include <cstdlib>
#include <mutex>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>          // compile-time check of the constants and signatures only; nothing links against it.  Without the RCCL
#define TDLO_HAVE_RCCL_H 1      // development headers the library still builds: RCCL is optional at run time, so it is at build time
#endif

namespace tdlo {

#ifdef TDLO_HAVE_RCCL_H
static_assert(kNcclFloat64 == (int)ncclFloat64 && kNcclSum == (int)ncclSum && kNcclMin == (int)ncclMin, "rccl.h enum values");
static_assert(sizeof(RcclApi::UniqueId) == sizeof(ncclUniqueId), "ncclUniqueId");
#endif

const RcclApi *rccl_api(const char *path_hint, std::string *why) {
    static RcclApi api;
    static std::string err;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (api.handle) return &api;
    const char *env = getenv("TDLO_RCCL_LIB");
    const char *cands[8]; int flags[8]; int n = 0;
    // an RCCL that is already mapped (e.g. by PyTorch) first: one process, one RCCL
    cands[n] = "librccl.so.1"; flags[n++] = RTLD_NOW | RTLD_NOLOAD;
    cands[n] = "librccl.so"; flags[n++] = RTLD_NOW | RTLD_NOLOAD;
    if (path_hint && path_hint[0]) { cands[n] = path_hint; flags[n++] = RTLD_NOW | RTLD_GLOBAL; }
    if (env && env[0]) { cands[n] = env; flags[n++] = RTLD_NOW | RTLD_GLOBAL; }
    cands[n] = "librccl.so.1"; flags[n++] = RTLD_NOW | RTLD_GLOBAL;
    cands[n] = "/opt/rocm/lib/librccl.so.1"; flags[n++] = RTLD_NOW | RTLD_GLOBAL;
    err.clear();
    for (int i = 0; i < n && !api.handle; ++i) {
        void *h = dlopen(cands[i], flags[i]);
        if (!h) { if (!(flags[i] & RTLD_NOLOAD)) { err += cands[i]; err += ": "; const char *e = dlerror(); err += e ? e : "?"; err += "; "; } continue; }
        RcclApi a;
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
        a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
        a.CommUserRank = (decltype(a.CommUserRank))dlsym(h, "ncclCommUserRank");
        a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
        if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString) { err += cands[i]; err += ": RCCL symbols missing; "; dlclose(h); continue; }
        a.handle = h; a.path = cands[i];
        api = a;
    }
    if (!api.handle) { if (why) *why = err.empty() ? "no librccl found" : err; return nullptr; }
    return &api;
}

}  // namespace tdlo
