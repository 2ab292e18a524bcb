/*
 * CPU oracle (plain C) for box_iou / filter_results / predict.
 * TEST INFRASTRUCTURE ONLY: built by oracle/Makefile into
 * oracle/liboracle.so and loaded by tests/, smoke() and bench.py's
 * cpu_baseline leg as the checker.  The product library never links it.
 *
 * Restates, operation by operation in IEEE float32 (build with
 * -ffp-contract=off, no fast-math):
 *   box_iou           /root/reference/utilis.py:298-323
 *   filter_img_class  /root/reference/utilis.py:355-390
 *   filter_results    /root/reference/utilis.py:393-465
 *   yolo_v3.predict   /root/reference/yolo_v3.py:205-262
 * Pinned by the golden vectors under tests/golden/ (captured from the
 * reference itself by tests/golden/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* torch.max / torch.min / clamp propagate NaN */
static float tmax(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
static float tmin(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }

/* utilis.py:298-323 */
float oracle_iou1(const float *p, const float *q)
{
    float ix1 = tmax(p[0], q[0]);
    float iy1 = tmax(p[1], q[1]);
    float ix2 = tmin(p[2], q[2]);
    float iy2 = tmin(p[3], q[3]);
    float iw = tmax((ix2 - ix1) + 1.0f, 0.0f);
    float ih = tmax((iy2 - iy1) + 1.0f, 0.0f);
    float inter = iw * ih;
    float a1 = ((p[2] - p[0]) + 1.0f) * ((p[3] - p[1]) + 1.0f);
    float a2 = ((q[2] - q[0]) + 1.0f) * ((q[3] - q[1]) + 1.0f);
    float uni = (a1 + a2) - inter;
    return inter / uni;
}

/* box1 is (1,4) when n1 == 1 (broadcast) or (m,4); box2 is (m,4) */
void oracle_box_iou(const float *box1, int n1, const float *box2, int m, float *out)
{
    for (int i = 0; i < m; ++i)
        out[i] = oracle_iou1(box1 + (n1 == 1 ? 0 : 4 * i), box2 + 4 * i);
}

typedef struct {
    int cls;
    float obj;
    int n;          /* original row index */
    float box[4];
    float score;
} cand_t;

static int cand_cmp(const void *a, const void *b)
{
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (x->cls != y->cls) return x->cls < y->cls ? -1 : 1;
    if (x->obj != y->obj) return x->obj > y->obj ? -1 : 1;   /* descending */
    return x->n < y->n ? -1 : (x->n > y->n ? 1 : 0);        /* stable */
}

/*
 * pred: (B, N, D) float32, D >= 5 + C.  out: room for B*N rows of 9 floats.
 * Returns K (rows written); 0 corresponds to the reference's int-0 sentinel.
 */
long oracle_filter_results(const float *pred, int B, int N, int D, int C,
                           float conf, float nms, float *out)
{
    long K = 0;
    cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (size_t)(N > 0 ? N : 1));
    unsigned char *dead = (unsigned char *)malloc((size_t)(N > 0 ? N : 1));
    for (int b = 0; b < B; ++b) {
        int m = 0;
        for (int n = 0; n < N; ++n) {
            const float *r = pred + ((size_t)b * N + n) * D;
            float obj = r[4];
            if (!(obj > conf) || !(obj > 0.0f)) continue;
            cand_t *c = &cand[m++];
            c->n = n;
            c->obj = obj;
            c->box[0] = r[0] - r[2] / 2.0f;
            c->box[1] = r[1] - r[3] / 2.0f;
            c->box[2] = r[0] + r[2] / 2.0f;
            c->box[3] = r[1] + r[3] / 2.0f;
            float best = r[5];
            int bi = 0;
            for (int j = 1; j < C; ++j) {
                float v = r[5 + j];
                if ((v > best) || ((v != v) && !(best != best))) { best = v; bi = j; }
            }
            c->score = best;
            c->cls = bi;
        }
        qsort(cand, (size_t)m, sizeof(cand_t), cand_cmp);
        int s = 0;
        while (s < m) {
            int e = s;
            while (e < m && cand[e].cls == cand[s].cls) ++e;
            memset(dead + s, 0, (size_t)(e - s));
            for (int i = s; i < e; ++i) {
                if (dead[i]) continue;
                float *o = out + K * 9;
                o[0] = (float)b;
                o[1] = cand[i].box[0]; o[2] = cand[i].box[1];
                o[3] = cand[i].box[2]; o[4] = cand[i].box[3];
                o[5] = cand[i].obj;    o[6] = cand[i].score;
                o[7] = (float)cand[i].cls;
                o[8] = cand[i].box[1];
                ++K;
                for (int j = i + 1; j < e; ++j) {
                    if (dead[j]) continue;
                    float iou = oracle_iou1(cand[i].box, cand[j].box);
                    if (!(iou < nms)) dead[j] = 1;
                }
            }
            s = e;
        }
    }
    free(cand);
    free(dead);
    return K;
}

/*
 * yolo_v3.py:205-262 for one head.  raw: (B, A*D, G, G) float32 (NCHW);
 * out: (B, G*G*A, D) rows ordered (gy*G + gx)*A + a.  anchors: A pairs,
 * already divided by the stride in float64 and rounded to float32.
 */
void oracle_decode_head(const float *raw, int B, int A, int D, int G,
                        const float *anchors_scaled, float stride, float *out)
{
    size_t plane = (size_t)G * G;
    for (int b = 0; b < B; ++b)
        for (int gy = 0; gy < G; ++gy)
            for (int gx = 0; gx < G; ++gx)
                for (int a = 0; a < A; ++a) {
                    float *o = out + (((size_t)b * plane + (size_t)gy * G + gx) * A + a) * D;
                    for (int d = 0; d < D; ++d) {
                        float t = raw[((size_t)b * A * D + (size_t)a * D + d) * plane + (size_t)gy * G + gx];
                        float v;
                        if (d == 0) v = (1.0f / (1.0f + expf(-t)) + (float)gx) * stride;
                        else if (d == 1) v = (1.0f / (1.0f + expf(-t)) + (float)gy) * stride;
                        else if (d == 2) v = (expf(t) * anchors_scaled[2 * a]) * stride;
                        else if (d == 3) v = (expf(t) * anchors_scaled[2 * a + 1]) * stride;
                        else v = 1.0f / (1.0f + expf(-t));
                        o[d] = v;
                    }
                }
}
