/* hip_loader.h -- dlopen() binding of libswimm_hip.so for the C host (include/swimm_hip.h). */
#ifndef SWIMM_HIP_LOADER_H_INCLUDED
#define SWIMM_HIP_LOADER_H_INCLUDED

#include "../../../include/swimm_hip.h"

typedef struct {
    void *handle;
    int (*abi_version)(void);
    const char *(*last_error)(void);
    int (*device_count)(void);
    int (*create)(int, swimm_hip_ctx **);
    void (*destroy)(swimm_hip_ctx *);
    int (*set_queries)(swimm_hip_ctx *, const char *, const uint16_t *, const uint32_t *, uint32_t, const char *, int, int);
    int (*add_chunk)(swimm_hip_ctx *, const char *, uint64_t, const uint16_t *, const uint32_t *, uint32_t, uint32_t, uint64_t);
    int (*add_sequences)(swimm_hip_ctx *, const uint16_t *, const char *, uint64_t, uint64_t);
    int (*clear_db)(swimm_hip_ctx *);
    int (*search)(swimm_hip_ctx *, int32_t *, uint64_t, double *);
    int (*search_topr)(swimm_hip_ctx *, uint32_t, uint64_t, int32_t *, int64_t *, double *);
    int (*last_stats)(swimm_hip_ctx *, double *, uint64_t *, uint64_t *, uint32_t *);
    int (*last_plan)(swimm_hip_ctx *, uint32_t, int *, int *, int *);
    int (*set_option)(swimm_hip_ctx *, const char *, int);
    int (*bind_host_thread)(int, int, char *, size_t);
} swimm_hip_api;

/* Looks for $SWIMM_HIP_LIB, then <dir of the executable>/../lib/libswimm_hip.so, then the loader path.
 * Returns 0 and fills *api, or non-zero with a message in err (no fallback of any kind). */
int swimm_hip_load(swimm_hip_api *api, char *err, unsigned long err_len);

#endif
