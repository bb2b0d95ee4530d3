/* affinity.c -- the CPU placement plan of affinity.h as functions of libswimm_host.so (bench.py ranks and the `swimm`
 * program's device threads bind themselves with it; tests/test_affinity.py drives the plan over a fake sysfs tree). */
#include "affinity.h"
#include "swimm_host.h"

int swimm_affinity_plan(const char *sysfs_root, const char *const *pci_bdf, int n_devices, int device, const int *allowed, int n_allowed,
                        int *out_cpus, int cap)
{
    return swimm_affinity_plan_impl(sysfs_root, pci_bdf, n_devices, device, allowed, n_allowed, out_cpus, cap);
}

int swimm_affinity_allowed(int *out, int cap) { return swimm_affinity_allowed_impl(out, cap); }

int swimm_affinity_apply(const int *cpus, int n) { return swimm_affinity_apply_impl(cpus, n); }
