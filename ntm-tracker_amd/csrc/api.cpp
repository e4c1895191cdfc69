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

// Compute units of the current device (hipDeviceAttributeMultiprocessorCount), cached per device: the cluster kernels
// need all their workgroups co-resident, one per CU, so their planners cap B * k at THIS number -- a partitioned
// (CPX / DPX) or smaller device gets a smaller k or the one-workgroup-per-sequence kernels, never a grid that cannot
// be resident.  256 (the MI355X in SPX mode) when no device can be queried: a launch would fail there anyway.
int ntk_device_cu_count() {
    static int cache[64];            // 0 = not queried yet (benign race: every writer stores the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    if (dev >= 0 && dev < 64 && __atomic_load_n(&cache[dev], __ATOMIC_RELAXED) > 0) return cache[dev];
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 256; }
    if (dev >= 0 && dev < 64) __atomic_store_n(&cache[dev], n, __ATOMIC_RELAXED);
    return n;
}
extern "C" int ntk_cu_count(void) { return ntk_device_cu_count(); }
