// comm.h -- RCCL, bound at run time.  The library does not link against RCCL: a single-GPU host never needs it, and a process
// that already holds a copy (PyTorch ships its own librccl.so.1) must not get a second one.  r3n_comm_* (r3n.hip) looks the
// handful of entry points up in the copy the process has loaded, else loads ROCm's.
#pragma once
#include <dlfcn.h>

#include <cstdlib>
#include <string>
#include <rccl/rccl.h>  // types and prototypes only

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    std::string error;

    bool load() {
        if (lib) return true;
        // R3N_RCCL_LIB: bind another implementation of the same entry points (tests/rccl_shim.cpp: ranks that share one GPU)
        if (const char *over = std::getenv("R3N_RCCL_LIB")) {
            lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
            if (!lib) { error = std::string("R3N_RCCL_LIB: ") + dlerror(); return false; }
        }
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            if (lib) break;
            lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);  // the copy already in the process (same soname: PyTorch's)
            if (lib) break;
        }
        if (!lib)
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (lib) break;
            }
        if (!lib) { error = std::string("RCCL not found: ") + dlerror(); return false; }
        auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p) error = std::string("RCCL symbol missing: ") + n; return p; };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
        ReduceScatter = reinterpret_cast<decltype(ReduceScatter)>(sym("ncclReduceScatter"));
        if (!error.empty()) { lib = nullptr; return false; }
        return true;
    }
};
inline Rccl &rccl() {
    static Rccl r;
    return r;
}
