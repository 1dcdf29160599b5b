/* Exhaustive check behind rand_component() of path_tracing_amd/csrc/pt_math.h: for every f32 x that an unsigned
 * 32-bit integer can convert to (x = 0 or 1 <= x <= 2^32), the three-operation sequence
 *     q0 = x * r;  q = fma(fma(-c, q0, x), r, q0)      with c = 3141.592653589793f, r = 1.0f / c
 * gives exactly the correctly rounded quotient x / c. Prints the number of mismatches; exit status 0 iff none.
 * Build: gcc -O2 -ffp-contract=off -mfma tools/verify_div_const.c -lm                                         */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

int main(void) {
    const float c = 3141.592653589793f;
    const float r = 1.0f / c;
    uint64_t bad = 0, n = 0;
    for (uint32_t b = 0; b <= 0x4f800000u; ++b) { /* 0x4f800000 = 2^32 */
        float x;
        memcpy(&x, &b, 4);
        if (b != 0 && x < 1.0f) continue;
        const float want = x / c;
        const float q0 = x * r;
        const float q = fmaf(fmaf(-c, q0, x), r, q0);
        if (memcmp(&q, &want, 4)) {
            if (bad < 5) printf("x=%a want=%a got=%a\n", x, want, q);
            ++bad;
        }
        ++n;
    }
    printf("checked %llu values, %llu mismatches\n", (unsigned long long)n, (unsigned long long)bad);
    return bad != 0;
}
