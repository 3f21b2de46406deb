// Would a hipGraph help the micro-batch chain?  A call of the latency path is two or three short dependent kernels on one stream
// (copy kernel -> latency form of K1 [-> K2]) followed by a stream synchronise.  This probe times exactly that shape with kernels that do
// a fixed, small amount of work (spin for `spin_us` each, like K1's 8-20 us), three ways: launched one by one, replayed as an instantiated
// graph captured from the same stream, and replayed with one kernel-node parameter update per call (what a real call would need: the
// batch size changes from call to call).  Host-to-host microseconds, p50 / p99 over 2000 calls.
//   hipcc -O3 --offload-arch=gfx950 profiles/graph_launch_probe.hip -o /tmp/graph_launch_probe && /tmp/graph_launch_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            std::printf("%s -> %s\n", #x, hipGetErrorString(e_));                     \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

__global__ void spin_kernel(uint32_t *out, uint32_t ticks, uint32_t tag)
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = tag;
}

// the same, but the last kernel of the chain announces completion itself: a sequence number stored to pinned host memory behind a
// system-scope fence; the host spins on that word instead of calling hipStreamSynchronize
__global__ void spin_flag_kernel(uint32_t *out, uint32_t ticks, uint32_t tag, volatile uint32_t *host_flag, uint32_t seq)
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = tag;
        if (host_flag) {
            __threadfence_system();
            *host_flag = seq;
        }
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void report(const char *name, int kernels, double spin_us, std::vector<double> &t)
{
    std::sort(t.begin(), t.end());
    std::printf("%-46s %d kernels x %4.1f us of work: p50 %6.1f us  p99 %6.1f us  (overhead over the work p50 %5.1f us)\n", name, kernels, spin_us,
                t[t.size() / 2], t[t.size() * 99 / 100], t[t.size() / 2] - kernels * spin_us);
}

int main(int argc, char **argv)
{
    // how the host waits inside hipStreamSynchronize: "spin" / "yield" / "block" set the device's schedule flag before anything else
    // touches the device (default: hipDeviceScheduleAuto); the runtime's environment switches are tried from the shell (sessions/s30.sh)
    if (argc > 1) {
        const std::string how = argv[1];
        const unsigned flag = how == "spin" ? hipDeviceScheduleSpin : how == "yield" ? hipDeviceScheduleYield : how == "block" ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto;
        CK(hipSetDeviceFlags(flag));
        std::printf("# hipSetDeviceFlags(%s)\n", how.c_str());
    }
    uint32_t *d = nullptr;
    CK(hipMalloc(&d, 256));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int calls = 2000;
    for (int kernels = 2; kernels <= 3; ++kernels) {
        for (double spin_us : {1.0, 8.0}) {
            const uint32_t ticks = (uint32_t)(spin_us * 100.0);
            std::vector<double> t;
            auto enqueue = [&](uint32_t tag) {
                for (int k = 0; k < kernels; ++k) hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(256), 0, st, d, ticks, tag + (uint32_t)k);
            };
            for (int i = 0; i < 200; ++i) { enqueue(i); CK(hipStreamSynchronize(st)); }
            for (int i = 0; i < calls; ++i) {
                const double a = now_us();
                enqueue((uint32_t)i);
                CK(hipStreamSynchronize(st));
                t.push_back(now_us() - a);
            }
            report("launched one by one + hipStreamSynchronize", kernels, spin_us, t);

            {
                uint32_t *flag = nullptr;
                CK(hipHostMalloc((void **)&flag, 64, hipHostMallocCoherent));
                *flag = 0;
                uint32_t seq = 0;
                t.clear();
                for (int i = 0; i < calls + 200; ++i) {
                    const double a = now_us();
                    ++seq;
                    for (int k = 0; k < kernels; ++k)
                        hipLaunchKernelGGL(spin_flag_kernel, dim3(8), dim3(256), 0, st, d, ticks, (uint32_t)(i + k), k == kernels - 1 ? flag : nullptr, seq);
                    while (*(volatile uint32_t *)flag != seq) {}
                    if (i >= 200) t.push_back(now_us() - a);
                }
                CK(hipStreamSynchronize(st));
                report("launched; host spins on a word the last kernel stores", kernels, spin_us, t);
                CK(hipHostFree(flag));
            }

            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            enqueue(7);
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int i = 0; i < 200; ++i) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
            t.clear();
            for (int i = 0; i < calls; ++i) {
                const double a = now_us();
                CK(hipGraphLaunch(ge, st));
                CK(hipStreamSynchronize(st));
                t.push_back(now_us() - a);
            }
            report("instantiated graph, replayed as captured", kernels, spin_us, t);

            size_t n_nodes = 0;
            CK(hipGraphGetNodes(g, nullptr, &n_nodes));
            std::vector<hipGraphNode_t> nodes(n_nodes);
            CK(hipGraphGetNodes(g, nodes.data(), &n_nodes));
            t.clear();
            for (int i = 0; i < calls + 200; ++i) {
                const double a = now_us();
                for (size_t k = 0; k < n_nodes; ++k) {  // a real call changes the batch size: every node gets new parameters
                    uint32_t tag = (uint32_t)i + (uint32_t)k;
                    uint32_t tk = ticks;
                    void *args[3] = {&d, &tk, &tag};
                    hipKernelNodeParams p{};
                    p.func = (void *)spin_kernel;
                    p.gridDim = dim3(8);
                    p.blockDim = dim3(256);
                    p.kernelParams = args;
                    CK(hipGraphExecKernelNodeSetParams(ge, nodes[k], &p));
                }
                CK(hipGraphLaunch(ge, st));
                CK(hipStreamSynchronize(st));
                if (i >= 200) t.push_back(now_us() - a);
            }
            report("graph + new parameters for every node per call", kernels, spin_us, t);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
