/* options.h -- command-line surface of the `swimm` program (reference: arguments.c:11-172,
 * arguments.h:9-37, swimm.c:4-7).  Same flags, same defaults where they still mean something. */
#ifndef SWIMM_OPTIONS_H_INCLUDED
#define SWIMM_OPTIONS_H_INCLUDED

#include <stdint.h>

#define MODE_CPU_ONLY 0      /* -m 0 : host CPU (explicit; never a fallback) */
#define MODE_GPU_ONLY 1      /* -m 1 : MI355X only (the reference's "Xeon Phi only") */
#define MODE_HYBRID 2        /* -m 2 : host CPU and MI355X concurrently, one work queue (hybrid_search) */

typedef struct {
    const char *op;               /* -S preprocess | search */
    const char *input_filename;   /* -i */
    const char *output_filename;  /* -o */
    const char *queries_filename; /* -q */
    const char *db_prefix;        /* -d */
    const char *submat_name;      /* -s (lower case) */
    int open_gap, extend_gap;     /* -g 10, -e 2 */
    int execution_mode;           /* -m, default 1 here (the reference defaults to 2) */
    int cpu_threads;              /* -c 4 */
    int num_gpus;                 /* -x 1 (number of accelerators) */
    int accel_threads;            /* -t 240, accepted and ignored */
    char profile;                 /* -p Q|S|A: query profile, score profile (sw_sp_kernel), adaptive (resolves to Q on gfx950) */
    int query_length_threshold;   /* -u 567: where the adaptive profile would consider the score profile */
    int vector_length;            /* -v 16|32 for -m 0; the GPU path lays the database out 128 wide */
    unsigned long top;            /* -r 10 */
    unsigned long max_chunk_size; /* -k 100663296 */
    int cpu_block_size;           /* -b 0 = default (60 for AVX2 width, 125 for SSE width; swimm.c:32-35) */
} swimm_options;

/* parses argv; on invalid input prints the argp message and exits(1) like the reference */
void swimm_parse_options(int argc, char **argv, swimm_options *o);

#endif
