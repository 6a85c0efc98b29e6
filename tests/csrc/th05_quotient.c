/* Exhaustive check of the float-only vote normalisation used by the S1 kernel for TH = 0.5
 * (patchperpix_amd/csrc/ppp_consensus_v2.hip, vote<.., TH05>): for EVERY float x in
 * [0.25, 2^22] the sequence
 *     d = x - 0.25f;  q0 = d * fl(4/3);  r = fma(-0.75f, q0, d);  y = fma(r, fl(4/3), q0)
 * must equal the reference's (float)(((double)x - 0.25) / 0.75)
 * (cuda/fillConsensusArray.cu:104-113: float product, double subtract and divide, float store).
 * Negative x follows by odd symmetry of every operation.  Prints the number of mismatches.
 * ppp_consensus_v3.hip uses the two-operation form  y = fma(d, fl(4/3), d * lo),
 * lo = fl(4/3 - fl(4/3)): checked over the same range (third number printed). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
int main(void) {
    const float c = 0x1.555556p+0f;
    const uint32_t lo = f2u(0.25f), hi = f2u(4194304.0f);
    const float cl = -0x1.555556p-25f;
    unsigned long long bad = 0, n = 0, bad2 = 0;
    if (cl != (float)(4.0 / 3.0 - (double)c)) bad2++;
    for (uint32_t u = lo; u <= hi; u++) {
        const float x = u2f(u);
        const float ref = (float)(((double)x - 0.25) / 0.75);
        const float d = x - 0.25f;
        const float q0 = d * c;
        const float r = fmaf(-0.75f, q0, d);
        const float y = fmaf(r, c, q0);
        n++;
        if (f2u(y) != f2u(ref)) bad++;
        if (f2u(fmaf(d, c, d * cl)) != f2u(ref)) bad2++;
        /* symmetric check on the negative side */
        const float xn = -x, dn = xn + 0.25f, q0n = dn * c, rn = fmaf(-0.75f, q0n, dn), yn = fmaf(rn, c, q0n);
        if (f2u(yn) != (f2u(ref) ^ 0x80000000u) && !(ref == 0.0f && yn == 0.0f)) bad++;
    }
    printf("%llu %llu %llu\n", n, bad, bad2);
    return bad != 0 || bad2 != 0;
}
