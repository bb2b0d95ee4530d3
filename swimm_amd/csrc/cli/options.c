/* options.c -- argp option table of `swimm` (flag surface of arguments.c:15-38). */
#include "options.h"

#include <argp.h>
#include <stdlib.h>
#include <string.h>

#include "../host/swimm_host.h"

const char *argp_program_version = "swimm " SWIMM_VERSION " (MI355X build)";

static char doc[] =
    "\nSWIMM (MI355X build): Smith-Waterman protein database search; the Xeon/Xeon Phi back-ends of the "
    "original are replaced by hand-written HIP kernels for AMD Instinct MI355X (gfx950)";

static struct argp_option options[] = {
    {0, 0, 0, 0, "SWIMM execution", 1},
    {0, 'S', "<string>", 0, "'preprocess' for database preprocessing, 'search' for database search. [REQUIRED]", 1},
    {0, 0, 0, 0, "preprocess", 2},
    {"input", 'i', "<string>", 0, "Input sequence filename (must be in FASTA format). [REQUIRED]", 2},
    {"output", 'o', "<string>", 0, "Output filename. [REQUIRED]", 2},
    {0, 0, 0, 0, "search", 3},
    {"query", 'q', "<string>", 0, "Input query sequence filename (must be in FASTA format). [REQUIRED]", 3},
    {"db", 'd', "<string>", 0, "Preprocessed database output filename. [REQUIRED]", 3},
    {"sm", 's', "<string>", 0, "Substitution matrix. Supported values: blosum45, blosum50, blosum62, blosum80, blosum90, pam30, pam70, pam250 (default: blosum62).", 3},
    {"gap_open", 'g', "<integer>", 0, "Gap open penalty (default: 10).", 3},
    {"gap_extend", 'e', "<integer>", 0, "Gap extend penalty (default: 2).", 3},
    {"execution_mode", 'm', "<integer>", 0, "Execution mode: 0 for host CPU only, 1 for MI355X only, 2 for concurrent host CPU and MI355X (default: 1).", 3},
    {"cpu_threads", 'c', "<integer>", 0, "Number of host threads (default: 4).", 3},
    {"num_gpus", 'x', "<integer>", 0, "Number of MI355X GPUs. Valid option only when execution mode is 1 (default: 1).", 3},
    {"mic_threads", 't', "<integer>", 0, "Accepted for compatibility; ignored (the GPU schedules its own wavefronts).", 3},
    {"mic_profile", 'p', "<char>", 0, "Profile technique on the GPU: 'Q' query profile, 'S' score profile, 'A' adaptive (default: resolves to the query profile, the faster one on gfx950 at every query length).", 3},
    {"query_length_threshold", 'u', "<integer>", 0, "Query length from which the adaptive profile would consider the score profile (default: 567).", 3},
    {"vector_length", 'v', "<integer>", 0, "Vector length for execution mode 0: 16 or 32 (default: 16). The GPU path always uses 128 sequences per wavefront.", 3},
    {"top", 'r', "<integer>", 0, "Number of scores to show (default: 10).", 3},
    {"max_chunk_size", 'k', "<integer>", 0, "Maximum chunk size in bytes. Valid option only when execution mode is 1 (default: 100663296).", 3},
    {"block_size", 'b', "<integer>", 0, "Host block size for execution mode 0 (default: 60 for vector length 32, 125 for 16).", 3},
    {0}};

static int parse_opt(int key, char *arg, struct argp_state *state)
{
    swimm_options *o = (swimm_options *)state->input;
    switch (key) {
    case 'S':
        if (strcmp(arg, "preprocess") != 0 && strcmp(arg, "search") != 0)
            argp_failure(state, 1, 0, "%s is not a valid option for execution.", arg);
        o->op = arg;
        break;
    case 'i': o->input_filename = arg; break;
    case 'o': o->output_filename = arg; break;
    case 'q': o->queries_filename = arg; break;
    case 'd': o->db_prefix = arg; break;
    case 's':
        if (!swimm_submat(arg)) argp_failure(state, 1, 0, "%s is not a valid option for substitution matrix.", arg);
        o->submat_name = arg;
        break;
    case 'g':
        o->open_gap = atoi(arg);
        if (o->open_gap < 0 || o->open_gap > 127) argp_failure(state, 1, 0, "%s is not a valid option for gap open penalty.", arg);
        break;
    case 'e':
        o->extend_gap = atoi(arg);
        if (o->extend_gap < 0 || o->extend_gap > 127) argp_failure(state, 1, 0, "%s is not a valid option for gap extend penalty.", arg);
        break;
    case 'm':
        o->execution_mode = atoi(arg);
        if (o->execution_mode < MODE_CPU_ONLY || o->execution_mode > MODE_HYBRID)
            argp_failure(state, 1, 0, "%d is not a valid option for execution mode.", o->execution_mode);
        break;
    case 'c':
        o->cpu_threads = atoi(arg);
        if (o->cpu_threads <= 0) argp_failure(state, 1, 0, "The number of host threads must be greater than 0.");
        break;
    case 'x':
        o->num_gpus = atoi(arg);
        if (o->num_gpus <= 0) argp_failure(state, 1, 0, "The number of GPUs must be greater than 0.");
        break;
    case 't': o->accel_threads = atoi(arg); break;
    case 'p':
        if (strcmp(arg, "Q") != 0 && strcmp(arg, "S") != 0 && strcmp(arg, "A") != 0)
            argp_failure(state, 1, 0, "%s is not a valid option for profile technique.", arg);
        o->profile = arg[0];
        break;
    case 'u':
        o->query_length_threshold = atoi(arg);
        if (o->query_length_threshold < 0 || o->query_length_threshold > 65535)      /* arguments.c:119-121 */
            argp_failure(state, 1, 0, "%s is not a valid option for query length threshold.", arg);
        break;
    case 'v':
        o->vector_length = atoi(arg);
        if (o->vector_length != 16 && o->vector_length != 32)
            argp_failure(state, 1, 0, "%d is not a valid option for vector length.", o->vector_length);
        break;
    case 'r': {
        long r = atol(arg);
        if (r <= 0) argp_failure(state, 1, 0, "The number of scores to show must be greater than 0.");
        o->top = (unsigned long)r;
        break;
    }
    case 'k': {
        long k = atol(arg);
        if (k <= 0) argp_failure(state, 1, 0, "The maximum chunk size must be greater than 0.");
        o->max_chunk_size = (unsigned long)k;
        break;
    }
    case 'b':
        o->cpu_block_size = atoi(arg);
        o->cpu_block_size = o->cpu_block_size / SWIMM_SEQ_LEN_MULT * SWIMM_SEQ_LEN_MULT;   /* arguments.c:152 */
        if (o->cpu_block_size <= 0) argp_failure(state, 1, 0, "The host block size must be at least %d.", SWIMM_SEQ_LEN_MULT);
        break;
    case ARGP_KEY_END:
        if (state->argc <= 1) argp_failure(state, 1, 0, "Missing options");
        if (!o->op) argp_failure(state, 1, 0, "SWIMM execution option is required");
        else if (strcmp(o->op, "preprocess") == 0) {
            if (!o->input_filename) argp_failure(state, 1, 0, "Input sequence filename is required");
            if (!o->output_filename) argp_failure(state, 1, 0, "Output filename is required");
        } else {
            if (!o->db_prefix) argp_failure(state, 1, 0, "Database filename is required");
            if (!o->queries_filename) argp_failure(state, 1, 0, "Query sequences filename is required");
            if (o->open_gap + o->extend_gap > 127)
                argp_failure(state, 1, 0, "Gap open + gap extend must not exceed 127.");
        }
        break;
    default:
        return ARGP_ERR_UNKNOWN;
    }
    return 0;
}

void swimm_parse_options(int argc, char **argv, swimm_options *o)
{
    memset(o, 0, sizeof *o);
    o->submat_name = "blosum62";
    o->open_gap = 10; o->extend_gap = 2;           /* arguments.h:19-20 */
    o->execution_mode = MODE_GPU_ONLY;
    o->cpu_threads = 4;                              /* arguments.h:14 */
    o->num_gpus = 1;                                 /* arguments.h:16 */
    o->accel_threads = 240;
    o->profile = 'A';
    o->query_length_threshold = 567;                 /* arguments.h:24 */
    o->vector_length = 16;                           /* arguments.h:10 */
    o->top = 10;                                     /* arguments.h:21 */
    o->max_chunk_size = 100663296;                   /* arguments.h:9 */
    struct argp argp = {options, parse_opt, 0, doc, 0, 0, 0};
    argp_parse(&argp, argc, argv, 0, 0, o);
}
