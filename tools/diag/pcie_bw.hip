// PCIe bandwidth of pinned host memory placed on each NUMA node (first touch from a CPU of that node), H2D and D2H,
// one stream and four concurrent streams.  hipcc -O2 pcie_bw.hip -o pcie_bw
#include <hip/hip_runtime.h>
#include <sched.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
static std::vector<int> node_cpus(int node)
{
    std::vector<int> v;
    char path[128]; snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r"); if (!f) return v;
    char buf[4096]; if (!fgets(buf, sizeof(buf), f)) { fclose(f); return v; } fclose(f);
    for (char *tok = strtok(buf, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a, b; if (sscanf(tok, "%d-%d", &a, &b) == 2) { for (int c = a; c <= b; ++c) v.push_back(c); } else if (sscanf(tok, "%d", &a) == 1) v.push_back(a);
    }
    return v;
}
int main()
{
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    printf("device %s pci %04x:%02x:%02x\n", pr.gcnArchName, pr.pciDomainID, pr.pciBusID, pr.pciDeviceID);
    { char p[256]; snprintf(p, sizeof(p), "cat /sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node 2>/dev/null", pr.pciDomainID, pr.pciBusID, pr.pciDeviceID); printf("gpu numa_node: "); fflush(stdout); (void)system(p); }
    const size_t bytes = 256u << 20;
    void *d; (void)hipMalloc(&d, bytes * 4);
    hipStream_t st[4]; for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int node = 0; node < 8; ++node) {
        std::vector<int> cpus = node_cpus(node);
        if (cpus.empty()) break;
        cpu_set_t set; CPU_ZERO(&set); for (int c : cpus) CPU_SET(c, &set);
        if (sched_setaffinity(0, sizeof(set), &set) != 0) { printf("node %d: cannot set affinity\n", node); continue; }
        void *h; if (hipHostMalloc(&h, bytes * 4, hipHostMallocPortable) != hipSuccess) { printf("node %d: alloc failed\n", node); continue; }
        memset(h, 1, bytes * 4);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int dir = 0; dir < 2; ++dir) {
            for (int ns = 1; ns <= 4; ns += 3) {
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    (void)hipDeviceSynchronize();
                    (void)hipEventRecord(e0, st[0]);
                    for (int k = 0; k < ns; ++k) {
                        char *hp = (char *)h + k * bytes, *dp = (char *)d + k * bytes;
                        if (dir == 0) (void)hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, st[k]); else (void)hipMemcpyAsync(hp, dp, bytes, hipMemcpyDeviceToHost, st[k]);
                    }
                    (void)hipDeviceSynchronize();
                    (void)hipEventRecord(e1, st[0]); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                printf("node %d (%zu cpus) %s %d stream(s): %.1f GB/s\n", node, cpus.size(), dir ? "D2H" : "H2D", ns, ns * bytes / best / 1e6);
            }
        }
        (void)hipHostFree(h);
    }
    return 0;
}
