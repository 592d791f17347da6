// Exhaustive comparison of the oracle's written-out glibc sinf / cosf (oracle/orb_ref.cpp, namespace glibc_flt32) with THIS machine's
// libm on every float of [0, 2 pi] - the arguments a key-point angle can take.  Built and run by tests/test_orb_oracle.py.
#include <gnu/libc-version.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
float orb_ref_glibc_sincosf(float y, int cosine);
int main(void) {
    const float hi = 6.2831860f;
    uint32_t uh;
    memcpy(&uh, &hi, 4);
    long bad_s = 0, bad_c = 0, n = 0, off_s = 0, off_c = 0;
#pragma omp parallel for reduction(+ : bad_s, bad_c, n, off_s, off_c) schedule(static)
    for (uint32_t u = 0; u <= uh; ++u) {
        float v;
        memcpy(&v, &u, 4);
        const float s = sinf(v), c = cosf(v);
        ++n;
        bad_s += orb_ref_glibc_sincosf(v, 0) != s;
        bad_c += orb_ref_glibc_sincosf(v, 1) != c;
        off_s += s != (float)sin((double)v);       // (how often libm is not the rounded double value)
        off_c += c != (float)cos((double)v);
    }
    printf("glibc %s: %ld floats in [0, 2 pi]: written-out sinf != libm on %ld, cosf on %ld; libm's sinf is off the rounded double sine on %ld, cosf on %ld\n",
           gnu_get_libc_version(), n, bad_s, bad_c, off_s, off_c);
    return (bad_s || bad_c) ? 1 : 0;
}
