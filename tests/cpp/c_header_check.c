/* include/glprover.h must be a plain C header (the boundary is a C ABI: cgo / ctypes / Rust bindgen read it as C): this file is
 * compiled with `gcc -std=c99 -pedantic -Wall -Werror` and linked against libglprover.so by tests/test_abi.py; it calls only host-side
 * entry points (no GPU). */
#include <stdio.h>
#include <string.h>

#include "glprover.h"

int main(void) {
    uint64_t zero[4] = {0, 0, 0, 0}, out[4];
    glp_challenger ch;
    glp_circuit cc;
    int32_t n = -1;
    memset(&cc, 0, sizeof cc);
    if (glp_abi_version() < 3) return 1;
    glp_two_to_one(zero, zero, out);
    glp_challenger_init(&ch);
    glp_challenger_observe(&ch, out, 4);
    (void)glp_challenger_get(&ch);
    if (glp_circuit_validate(&cc) >= 0) return 2;  /* an all-zero description is not a circuit */
    if (glp_ctx_memory(NULL, NULL, NULL) != GLP_ERR_INVALID) return 3;
    (void)glp_device_count(&n); /* 0 devices and an error status on a CPU box: both fine here */
    printf("c header ok: two_to_one(0,0)[0] = %llu, %s\n", (unsigned long long)out[0], glp_strerror(GLP_ERR_OOM));
    return 0;
}
