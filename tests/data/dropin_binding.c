/* Compile-only check (tests/test_abi_symbols.py): include/swimm_hip.h is plain C, and the whole-call entry point
 * accepts exactly the argument types swimm.c holds when it calls mic_search_knc_ap_multiple_chunks (swimm.c:88-90,
 * MICsearch.h:35-38) -- the binding shown in INTEGRATION.md section 2. */
#include "swimm_hip.h"

typedef int (*search_chunks_fn)(const char *, const unsigned short *, unsigned int, const unsigned int *,
                                unsigned long, char **, unsigned int, const unsigned int *, unsigned short **, unsigned int **,
                                const unsigned long *, const char *, int, int, int, unsigned int, int *, double *);

int bind_and_call(char *query_sequences, unsigned short *m, unsigned int query_sequences_count, unsigned int *query_sequences_disp,
                  unsigned long vect_sequences_db_count, char **chunk_vect_sequences_db, unsigned int chunk_count,
                  unsigned int *chunk_vect_sequences_db_count, unsigned short **chunk_vect_sequences_db_lengths,
                  unsigned int **chunk_vect_sequences_db_disp, unsigned long *chunk_vD, char *submat, int open_gap, int extend_gap,
                  int num_mics, int vector_length, int *scores, double *workTime)
{
    search_chunks_fn search = swimm_hip_search_chunks;   /* incompatible types are an error under -Werror */
    return search(query_sequences, m, query_sequences_count, query_sequences_disp, vect_sequences_db_count,
                  chunk_vect_sequences_db, chunk_count, chunk_vect_sequences_db_count, chunk_vect_sequences_db_lengths,
                  chunk_vect_sequences_db_disp, chunk_vD, submat, open_gap, extend_gap, num_mics, (unsigned int)vector_length,
                  scores, workTime);
}
