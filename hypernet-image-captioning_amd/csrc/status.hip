// Sticky device-side failure word (include/caphn.h: caphn_device_error).  One int per device in pinned, device-mapped host
// memory: a kernel that detects a failure it cannot recover from (today: a pair recurrent kernel whose partner workgroup
// never answered) stores a code with system scope; the host reads the word with a plain load at every libcaphn call that
// launches work (caphn_launch_status) -- no synchronisation, no device read-back on the training path.
#include "common.h"
#include <mutex>

long long g_tune_xch_timeout = 100000000ll;      // 1 s at 100 MHz (caphn_tune key 22: microseconds)

namespace {
constexpr int MAX_DEVICES = 64;
int* g_word[MAX_DEVICES];            // host address == device address (unified addressing of hipHostMalloc memory)
std::mutex g_mu;
inline int here() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return -1;
    return dev;
}
}  // namespace

int* caphn_errword() {
    const int dev = here();
    if (dev < 0) return nullptr;
    if (g_word[dev]) return g_word[dev];
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_word[dev]) return g_word[dev];
    void* p = nullptr;
    // (first use is a first launch of the pair kernels: run one eager step before capturing a graph, as for the side streams)
    if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *static_cast<volatile int*>(p) = 0;
    g_word[dev] = static_cast<int*>(p);
    return g_word[dev];
}

int caphn_sticky_error() {
    const int dev = here();
    if (dev < 0 || !g_word[dev]) return CAPHN_OK;
    return *static_cast<volatile int*>(g_word[dev]) != 0 ? CAPHN_ETIMEOUT : CAPHN_OK;
}

extern "C" int caphn_device_error(int clear) {
    const int dev = here();
    if (dev < 0 || !g_word[dev]) return CAPHN_OK;
    volatile int* w = g_word[dev];
    const int v = *w;
    if (clear) *w = 0;
    return v != 0 ? CAPHN_ETIMEOUT : CAPHN_OK;
}
