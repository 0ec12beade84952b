/* CPU ORACLE (test infrastructure, not product code): the vector quantiser's arithmetic in the exact
 * operation order of paintmind_amd/csrc/vq.hip, so that token indices can be compared bit-exactly.
 * Follows reference stage1/quantize.py:18-30 (l2norm :5-6, distance :24-26, argmin :28).
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC vq_ref.c -o _build/libvq_ref.so -lm   (oracle/build.sh)
 * Every multiply-add is an explicit fmaf (single rounding), sqrt/divide are IEEE. */
#include <math.h>
#include <stdint.h>

void vq_ref_prepare(const float* w, float* en, float* sq, int V, int E) {
    for (int j = 0; j < V; ++j) {
        float ss = 0.f;
        for (int k = 0; k < E; ++k) ss = fmaf(w[(long)j * E + k], w[(long)j * E + k], ss);
        const float den = fmaxf(sqrtf(ss), 1e-12f);
        float s2 = 0.f;
        for (int k = 0; k < E; ++k) {
            const float v = w[(long)j * E + k] / den;
            en[(long)j * E + k] = v;
            s2 = fmaf(v, v, s2);
        }
        sq[j] = s2;
    }
}

/* z [M,E] -> idx [M], zn [M,E] (normalised rows), dmin [M], gap [M] = second-best distance - best */
void vq_ref_quantize(const float* z, const float* en, const float* sq, int M, int V, int E, int64_t* idx,
                     float* zn_out, float* dmin, float* gap) {
    float zn[64];
    for (int m = 0; m < M; ++m) {
        float ss = 0.f;
        for (int k = 0; k < E; ++k) ss = fmaf(z[(long)m * E + k], z[(long)m * E + k], ss);
        const float den = fmaxf(sqrtf(ss), 1e-12f);
        float sz = 0.f;
        for (int k = 0; k < E; ++k) {
            zn[k] = z[(long)m * E + k] / den;
            sz = fmaf(zn[k], zn[k], sz);
        }
        float best = INFINITY, second = INFINITY;
        int bi = 0;
        for (int j = 0; j < V; ++j) {
            float dot = 0.f;
            for (int k = 0; k < E; ++k) dot = fmaf(zn[k], en[(long)j * E + k], dot);
            const float d = (sz + sq[j]) - 2.0f * dot;
            if (d < best) { second = best; best = d; bi = j; }
            else if (d < second) second = d;
        }
        idx[m] = bi;
        if (dmin) dmin[m] = best;
        if (gap) gap[m] = second - best;
        if (zn_out) for (int k = 0; k < E; ++k) zn_out[(long)m * E + k] = zn[k];
    }
}
