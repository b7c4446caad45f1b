/* A plain C99 caller of the boundary: links libdeacon_hip.so, needs no GPU for what it calls.
 * (tests/test_abi.py::test_c_caller_links_and_runs builds and runs it.)  Exit code 0 = every check held. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "deacon_hip.h"

int main(void) {
    const char *v = dcn_version();
    if (!v || strncmp(v, "deacon-hip ", 11) != 0) return 1;

    /* dcn_pack_ascii: PackedSeqVec::from_ascii's code (c >> 1) & 3 and the invalid-base mask, 32-base groups */
    const char *seq = "ACGTacgtNNRYACGTACGTACGTACGTACGTACGTA"; /* 37 bases: two groups */
    const uint64_t n = (uint64_t)strlen(seq);
    uint32_t packed[4] = {0, 0, 0, 0}, mask[2] = {0, 0};
    uint32_t nl = 7;
    if (dcn_pack_ascii((const uint8_t *)seq, n, packed, mask, &nl) != DCN_OK || nl != 0) return 2;
    for (uint64_t i = 0; i < n; ++i) {
        unsigned code = (packed[i / 16] >> (2 * (i % 16))) & 3u;
        unsigned bad = (mask[i / 32] >> (i % 32)) & 1u;
        char c = seq[i], u = (char)(c & ~0x20);
        if (code != (((unsigned)c >> 1) & 3u)) return 3;
        if (bad != !(u == 'A' || u == 'C' || u == 'G' || u == 'T')) return 4;
    }
    /* argument errors are reported, never aborted on; the message is thread-local text */
    if (dcn_pack_ascii(NULL, 5, packed, mask, NULL) != DCN_ERR_ARG || strlen(dcn_last_error()) == 0) return 5;
    /* a line end in the input is reported: the packed entry points cannot strip it (src/filter_common.rs:229) */
    if (dcn_pack_ascii((const uint8_t *)"ACGT\nACGT\n", 10, packed, mask, &nl) != DCN_OK || nl != 1) return 9;
    dcn_index *idx = NULL;
    int rc = dcn_index_from_keys(NULL, 3, 31, 15, 0, &idx); /* keys == NULL with n > 0 */
    if (rc != DCN_ERR_ARG || idx != NULL) return 6;
    rc = dcn_index_from_file("/nonexistent/file.idx", 0, &idx);
    if (rc == DCN_OK || idx != NULL) return 7;
    int ndev = -1;
    rc = dcn_device_count(&ndev); /* DCN_OK with a GPU, DCN_ERR_HIP without: either way no abort and a defined count */
    if (ndev < 0) return 8;
    /* the parity-pinning switch: argument errors are refused, a valid setting reads back, the default is (1, 16, 0) */
    uint32_t rot = 0, bits = 0, comb = 9;
    if (dcn_get_minimizer_variant(&rot, &bits, &comb) != DCN_OK || rot != 1 || bits != 16 || comb != 0) return 10;
    if (dcn_set_minimizer_variant(0, 16, 0) != DCN_ERR_ARG || dcn_set_minimizer_variant(1, 24, 0) != DCN_ERR_ARG ||
        dcn_set_minimizer_variant(1, 16, 2) != DCN_ERR_ARG)
        return 11;
    if (dcn_set_minimizer_variant(7, 32, 1) != DCN_OK || dcn_get_minimizer_variant(&rot, &bits, &comb) != DCN_OK ||
        rot != 7 || bits != 32 || comb != 1 || dcn_set_minimizer_variant(1, 16, 0) != DCN_OK)
        return 12;
    /* ABI version: a binding refuses a library of another major */
    uint32_t major = 0, minor = 0;
    if (dcn_abi_version(&major, &minor) != DCN_OK || major != DCN_ABI_MAJOR || minor < DCN_ABI_MINOR) return 14;
    if (dcn_abi_version(NULL, &minor) != DCN_ERR_ARG) return 15;
    /* a non-default rule refuses windows its generic kernel cannot hold, when the index is made (not at the first filter call) */
    if (dcn_set_minimizer_variant(7, 16, 0) != DCN_OK) return 16;
    {
        const uint64_t one_key = 1;
        rc = dcn_index_from_keys(&one_key, 1, 31, 201, 0, &idx);
        if (rc != DCN_ERR_ARG || idx != NULL || strstr(dcn_last_error(), "w <= 128") == NULL) return 17;
    }
    if (dcn_set_minimizer_variant(1, 16, 0) != DCN_OK) return 18;
    /* the RCCL communicator: argument errors without touching a GPU */
    {
        dcn_comm *comm = NULL;
        uint8_t id[DCN_COMM_ID_BYTES];
        uint64_t counters[DCN_N_STATS];
        memset(id, 0, sizeof id);
        if (dcn_comm_create(NULL, 1, 0, 0, &comm) != DCN_ERR_ARG || dcn_comm_create(id, 2, 2, 0, &comm) != DCN_ERR_ARG || comm != NULL) return 19;
        if (dcn_comm_unique_id(NULL) != DCN_ERR_ARG || dcn_stats_allreduce_rccl(NULL, NULL, 0, counters) != DCN_ERR_ARG) return 20;
        dcn_comm_destroy(NULL);
    }
    double rate = -1.0;
    if (dcn_index_probe_ceiling(NULL, NULL, 10, 1, &rate) != DCN_ERR_ARG) return 13;
    printf("%s devices=%d rc=%d\n", v, ndev, rc);
    return 0;
}
