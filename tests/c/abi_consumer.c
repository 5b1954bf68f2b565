/* A plain C99 consumer of the C ABI (include/lpi_hip.h): no Python, no torch, no C++ — what a cgo / JNI / FFI binding of another host language sees.
 * Built and run by tests/test_capi_symbols.py::test_plain_c_consumer on a CPU-only machine: it makes no device call (lpi_version, the *_supported
 * predicates, the host-side BPE entry points only need the library to load). */
#include <stdio.h>
#include "lpi_hip.h"

int main(void) {
    int v = lpi_version();
    int rows_ok = lpi_gemm_nt_rows_supported(LPI_BF16, 256, 768, 3072);       /* a few-row GEMM shape of the step */
    int rows_bad = lpi_gemm_nt_rows_supported(LPI_BF16, 250, 768, 3072);      /* rows that are not whole 32-row tiles */
    int spool_ok = lpi_spool_attn_supported(213, 12, 768);                     /* ViT-B/16's last block without K and V */
    int spool_long = lpi_spool_attn_supported(593, 16, 1024);                  /* ViT-L/14@336px: the score table does not fit */
    printf("lpi_version=%d rows_ok=%d rows_bad=%d spool_ok=%d spool_long=%d\n", v, rows_ok, rows_bad, spool_ok, spool_long);
    return (v > 0 && rows_ok == 1 && rows_bad == 0 && spool_ok == 1 && spool_long == 0) ? 0 : 1;
}
