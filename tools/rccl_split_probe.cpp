// rccl_split_probe.cpp -- does this RCCL build accept ncclCommSplit (the slab path's second communicator)?
// One rank only (a one-GPU box cannot host two): API availability and a self send/recv on each communicator.
// hipcc -o /tmp/rccl_split_probe tools/rccl_split_probe.cpp -lrccl && /tmp/rccl_split_probe
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#define CK(x) do { ncclResult_t r_ = (x); printf("%-60s %s\n", #x, ncclGetErrorString(r_)); if (r_ != ncclSuccess) return 1; } while (0)
int main()
{
    ncclUniqueId id;
    ncclComm_t a, b;
    CK(ncclGetUniqueId(&id));
    CK(ncclCommInitRank(&a, 1, id, 0));
    CK(ncclCommSplit(a, 0, 0, &b, nullptr));
    double *p;
    hipMalloc(&p, 1 << 20);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    CK(ncclAllGather(p, p, 16, ncclDouble, a, s1));
    CK(ncclBroadcast(p, p, 16, ncclDouble, 0, b, s2));
    printf("sync %d %d\n", (int)hipStreamSynchronize(s1), (int)hipStreamSynchronize(s2));
    CK(ncclCommDestroy(b));
    CK(ncclCommDestroy(a));
    return 0;
}
