/*
 * knn_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Scalar C restatement of the reference's simple-knn (distCUDA2): mean of the three
 * smallest squared distances from each point to the other points.
 *   KNN/ = dgmesh/submodules/simple-knn/
 * Follows KNN/simple_knn.cu:45-61 (Morton code), :78-117 (box AABBs), :119-183 (distBoxPoint,
 * updateKBest<3>, boxMeanDist), :185-221 (driver; note the min/max reductions both start
 * from init = {0,0,0}, :191-199, so the bounding box always contains the origin).
 *
 * The box/reject heuristic only prunes: every box that can hold one of the 3 nearest
 * neighbours is brute-forced, so the result equals the exact 3-NN mean of the fp32 values
 * d2 = dx*dx + dy*dy + dz*dz (left to right, no FMA: build with -ffp-contract=off), and
 * orc_knn_brute below is the O(P^2) statement of that definition used to pin this file.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BOX_SIZE 1024

static uint32_t f2u_sat(float f) {
    if (f != f) return 0u;
    if (f >= 4294967296.0f) return 4294967295u;
    if (f <= 0.0f) return 0u;
    return (uint32_t)f;
}

/* KNN/simple_knn.cu:45-52 */
static uint32_t prepMorton(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
/* KNN/simple_knn.cu:54-61 */
static uint32_t coord2Morton(const float* c, const float* mn, const float* mx) {
    uint32_t x = prepMorton(f2u_sat(((c[0] - mn[0]) / (mx[0] - mn[0])) * ((1 << 10) - 1)));
    uint32_t y = prepMorton(f2u_sat(((c[1] - mn[1]) / (mx[1] - mn[1])) * ((1 << 10) - 1)));
    uint32_t z = prepMorton(f2u_sat(((c[2] - mn[2]) / (mx[2] - mn[2])) * ((1 << 10) - 1)));
    return x | (y << 1) | (z << 2);
}

/* KNN/simple_knn.cu:131-145 (updateKBest<3>) */
static void updateKBest3(const float* ref, const float* p, float* knn) {
    float d[3] = {p[0] - ref[0], p[1] - ref[1], p[2] - ref[2]};
    float dist = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    for (int j = 0; j < 3; j++) {
        if (knn[j] > dist) {
            float t = knn[j];
            knn[j] = dist;
            dist = t;
        }
    }
}

/* KNN/simple_knn.cu:119-129 */
static float distBoxPoint(const float* bmin, const float* bmax, const float* p) {
    float diff[3] = {0, 0, 0};
    for (int a = 0; a < 3; a++)
        if (p[a] < bmin[a] || p[a] > bmax[a]) diff[a] = fminf(fabsf(p[a] - bmin[a]), fabsf(p[a] - bmax[a]));
    return diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
}

void orc_knn(int P, const float* points, float* meanDists) {
    if (P <= 0) return;
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (int i = 0; i < P; i++)
        for (int a = 0; a < 3; a++) {
            mn[a] = fminf(mn[a], points[3 * i + a]);
            mx[a] = fmaxf(mx[a], points[3 * i + a]);
        }
    uint32_t* codes = (uint32_t*)malloc(sizeof(uint32_t) * P);
    uint32_t* idx = (uint32_t*)malloc(sizeof(uint32_t) * P);
    uint32_t* codes2 = (uint32_t*)malloc(sizeof(uint32_t) * P);
    uint32_t* idx2 = (uint32_t*)malloc(sizeof(uint32_t) * P);
    for (int i = 0; i < P; i++) {
        codes[i] = coord2Morton(points + 3 * i, mn, mx);
        idx[i] = (uint32_t)i;
    }
    /* cub::DeviceRadixSort::SortPairs == stable LSD radix sort (KNN/simple_knn.cu:210-213) */
    for (int pass = 0; pass < 4; pass++) {
        size_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        int sh = 8 * pass;
        for (int i = 0; i < P; i++) cnt[((codes[i] >> sh) & 0xff) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (int i = 0; i < P; i++) {
            size_t dst = cnt[(codes[i] >> sh) & 0xff]++;
            codes2[dst] = codes[i];
            idx2[dst] = idx[i];
        }
        uint32_t* t = codes;
        codes = codes2;
        codes2 = t;
        t = idx;
        idx = idx2;
        idx2 = t;
    }
    int nb = (P + BOX_SIZE - 1) / BOX_SIZE;
    float* bmin = (float*)malloc(sizeof(float) * 3 * nb);
    float* bmax = (float*)malloc(sizeof(float) * 3 * nb);
    for (int b = 0; b < nb; b++) {
        for (int a = 0; a < 3; a++) {
            bmin[3 * b + a] = FLT_MAX;
            bmax[3 * b + a] = -FLT_MAX;
        }
        for (int i = b * BOX_SIZE; i < P && i < (b + 1) * BOX_SIZE; i++)
            for (int a = 0; a < 3; a++) {
                bmin[3 * b + a] = fminf(bmin[3 * b + a], points[3 * idx[i] + a]);
                bmax[3 * b + a] = fmaxf(bmax[3 * b + a], points[3 * idx[i] + a]);
            }
    }
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < P; i++) {
        const float* point = points + 3 * idx[i];
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        int lo = i - 3 > 0 ? i - 3 : 0, hi = i + 3 < P - 1 ? i + 3 : P - 1;
        for (int j = lo; j <= hi; j++) {
            if (j == i) continue;
            updateKBest3(point, points + 3 * idx[j], best);
        }
        float reject = best[2];
        best[0] = best[1] = best[2] = FLT_MAX;
        for (int b = 0; b < nb; b++) {
            float dist = distBoxPoint(bmin + 3 * b, bmax + 3 * b, point);
            if (dist > reject || dist > best[2]) continue;
            int end = (b + 1) * BOX_SIZE < P ? (b + 1) * BOX_SIZE : P;
            for (int j = b * BOX_SIZE; j < end; j++) {
                if (j == i) continue;
                updateKBest3(point, points + 3 * idx[j], best);
            }
        }
        meanDists[idx[i]] = (best[0] + best[1] + best[2]) / 3.0f;
    }
    free(codes);
    free(idx);
    free(codes2);
    free(idx2);
    free(bmin);
    free(bmax);
}

/* The definition orc_knn must reproduce bit for bit: exhaustive 3 smallest fp32 d2 per point. */
void orc_knn_brute(int P, const float* points, float* meanDists) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            updateKBest3(points + 3 * i, points + 3 * j, best);
        }
        meanDists[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}
