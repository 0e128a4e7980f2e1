/* TEST INFRASTRUCTURE ONLY -- see er_group_oracle.h. */
#include "er_group_oracle.h"
#include "er_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

void ero_calc_color(const uint8_t *mask_plane, int mstride, int bx, int by, int bw, int bh,
                    const uint8_t *color_img, int cstride, double out[3])
{
    const uint8_t *roi = mask_plane + (size_t)by * mstride + bx;
    const int th = ero_otsu_threshold(roi, mstride, bw, bh, 1);        /* threshold(255-img, img, 128, 255, THRESH_OTSU) */
    int    count = 0;
    double color1 = 0, color2 = 0, color3 = 0;
    for (int i = 0; i < bh; ++i) {
        const uint8_t *ptr = roi + (size_t)i * mstride;
        const uint8_t *color_ptr = color_img + (size_t)i * cstride;    /* row i of the whole image (:1404) */
        for (int j = 0, k = 0; j < bw; ++j, k += 3)
            if ((255 - ptr[j]) > th) {
                ++count;
                color1 += color_ptr[k];
                color2 += color_ptr[k + 1];
                color3 += color_ptr[k + 2];
            }
    }
    out[0] = color1 / count;
    out[1] = color2 / count;
    out[2] = color3 / count;
}

int ero_er_track(ero_er *ers, int n, int *order)
{
    for (int i = 0; i < n; ++i) { ers[i].cx = ers[i].x + ers[i].w / 2; ers[i].cy = ers[i].y + ers[i].h / 2; }
    int na = 0;
    for (int i = 0; i < n; ++i) if (ers[i].cls == 1) order[na++] = i;
    uint8_t *tracked = (uint8_t *)calloc((size_t)n + 1, 1);
    for (int i = 0; i < na; ++i) {
        const ero_er *s = &ers[order[i]];
        for (int k = 0; k < n; ++k) {
            if (ers[k].cls != 2 || tracked[k]) continue;
            const ero_er *w = &ers[k];
            if (abs(s->cx - w->cx) + abs(s->cy - w->cy) < imax(s->w, s->h) << 1 &&
                abs(s->h - w->h) < imin(s->h, w->h) &&
                abs(s->w - w->w) < (s->w + w->w) >> 1 &&
                fabs(s->color1 - w->color1) < 25 &&
                fabs(s->color2 - w->color2) < 25 &&
                fabs(s->color3 - w->color3) < 25 &&
                abs(s->area - w->area) < imin(s->area, w->area) * 3) {
                tracked[k] = 1;
                order[na++] = k;
            }
        }
    }
    free(tracked);
    return na;
}
