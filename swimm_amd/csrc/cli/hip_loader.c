/* hip_loader.c -- see hip_loader.h */
#define _GNU_SOURCE
#include "hip_loader.h"

#include <dlfcn.h>
#include <libgen.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define BIND(field, name)                                                                   \
    do {                                                                                    \
        *(void **)(&api->field) = dlsym(api->handle, name);                                 \
        if (!api->field) { snprintf(err, err_len, "SWIMM: %s lacks symbol %s", path, name); return 1; } \
    } while (0)

int swimm_hip_load(swimm_hip_api *api, char *err, unsigned long err_len)
{
    memset(api, 0, sizeof *api);
    char path[PATH_MAX + 64] = "";
    const char *env = getenv("SWIMM_HIP_LIB");
    if (env && *env) {
        snprintf(path, sizeof path, "%s", env);
        api->handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    } else {
        char exe[PATH_MAX];
        ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
        if (n > 0) {
            exe[n] = 0;
            snprintf(path, sizeof path, "%s/../lib/libswimm_hip.so", dirname(exe));
            api->handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        }
        if (!api->handle) {
            snprintf(path, sizeof path, "libswimm_hip.so");
            api->handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        }
    }
    if (!api->handle) {
        snprintf(err, err_len, "SWIMM: cannot load the MI355X back-end (%s). Build it with `make -C swimm_amd/csrc` or set SWIMM_HIP_LIB.", dlerror());
        return 1;
    }
    BIND(abi_version, "swimm_hip_abi_version");
    BIND(last_error, "swimm_hip_last_error");
    BIND(device_count, "swimm_hip_device_count");
    BIND(create, "swimm_hip_create");
    BIND(destroy, "swimm_hip_destroy");
    BIND(set_queries, "swimm_hip_set_queries");
    BIND(add_chunk, "swimm_hip_add_chunk");
    BIND(add_sequences, "swimm_hip_add_sequences");
    BIND(clear_db, "swimm_hip_clear_db");
    BIND(search, "swimm_hip_search");
    BIND(search_topr, "swimm_hip_search_topr");
    BIND(last_stats, "swimm_hip_last_stats");
    BIND(last_plan, "swimm_hip_last_plan");
    BIND(set_option, "swimm_hip_set_option");
    BIND(bind_host_thread, "swimm_hip_bind_host_thread");
    if (api->abi_version() != SWIMM_HIP_ABI_VERSION) {
        snprintf(err, err_len, "SWIMM: %s has ABI version %d, this program needs %d", path, api->abi_version(), SWIMM_HIP_ABI_VERSION);
        return 1;
    }
    return 0;
}
