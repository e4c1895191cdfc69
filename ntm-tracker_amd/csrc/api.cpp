// Error reporting, version and the per-device attribute cache of libntmtrack_hip.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/ntmtrack.h"

static thread_local char g_err[512] = "";

void ntk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int ntk_version(void) { return 100; }
extern "C" const char* ntk_last_error(void) { return g_err; }

// Compute units the cooperative (cluster) kernels may count on, cached per device: those kernels need all their workgroups
// co-resident, one per CU, so their planners cap B * k at THIS number -- a partitioned (CPX / DPX) or smaller device gets a
// smaller k or the one-workgroup-per-sequence kernels, never a grid that cannot be resident.  The number is
// hipDeviceAttributeMultiprocessorCount, which assumes the process has the device to ITSELF: a CU mask (HSA_CU_MASK /
// ROC_GLOBAL_CU_MASK) or another process's resident kernels are invisible to it, and a grid of exactly that many workgroups
// would then spin until its bounded waits abort, every step.  NTK_DNC_CU_BUDGET=<n> (read once) overrides the count for such
// deployments.  A failed query returns 0: every planner then refuses the cluster forms (B * k <= 0 never holds) and the
// one-workgroup-per-sequence kernels run -- nothing assumes 256.
#include <stdlib.h>
int ntk_device_cu_count() {
    static int cache[64];            // 0 = not queried yet (benign race: every writer stores the same value)
    static int budget = -1;          // -1 = environment not read yet
    if (__atomic_load_n(&budget, __ATOMIC_RELAXED) < 0) {
        const char* e = getenv("NTK_DNC_CU_BUDGET");
        int b = 0;
        if (e && *e) { b = atoi(e); if (b < 0) b = 0; }
        __atomic_store_n(&budget, b, __ATOMIC_RELAXED);
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int n = (dev >= 0 && dev < 64) ? __atomic_load_n(&cache[dev], __ATOMIC_RELAXED) : 0;
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 0; }
        if (dev >= 0 && dev < 64) __atomic_store_n(&cache[dev], n, __ATOMIC_RELAXED);
    }
    const int b = __atomic_load_n(&budget, __ATOMIC_RELAXED);
    return (b > 0 && b < n) ? b : n;
}
extern "C" int ntk_cu_count(void) { return ntk_device_cu_count(); }
