// Cost model of a BIT-PLANE form of the tile kernel, on real planes:  g++ -O2 tools/sim_bitplane.cpp -o /tmp/sim_bp && /tmp/sim_bp text_0.lev 1920 1080
// (plane files from tools/sim_tile_dump.py: one byte per pixel = quantised level, 255 = wall).
//
// The form that is counted: one lane per tile ROW (64 pixels = one 64-bit mask per level), 32 lanes per tile, TWO tiles per wave -- 2 x 2048 pixels per wave,
// four times the piece kernel's 512.  For every level t that occurs in the tile, bottom-up:
//   masks   M_t = pixels of level <= t, from five bit planes of the level:                    ~20 vector instructions per level (64-bit, per row-lane)
//   runs    the runs of M_t in the lane's row are its units; a lane walks ITS runs one after another, the wave as many rounds as its fullest row has runs;
//           per run and round: isolate the run (s_ff1-like bit tricks on 64 bits), test it against the row above / below for overlaps, one union per
//           overlapping run pair, and "does it hold a pixel of level exactly t" (a node is born) + area / box updates:          ~30 vector instructions per round
//   (the runs of M_t are re-walked at every level because they GROW and merge with t: the labelling of level t - 1 cannot be kept per run)
// The model counts rounds = sum over the levels present of (max over the rows of a tile pair of the number of runs of M_t in the row), and prices
//   instructions per wave = levels * 20 + rounds * 30,  per 512 pixels = that / 8
// against the piece kernel's measured 1320 vector (2389 all) instructions per wave of 512 pixels.  It is a LOWER bound of the form: unions are counted as one
// instruction-group each although they are find loops, and nothing is charged for the statistics, the fold, the export or the seam map.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv)
{
    if (argc < 4) { puts("usage: sim_bp plane.lev W H"); return 1; }
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    std::vector<uint8_t> lev((size_t)W * H);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(lev.data(), 1, lev.size(), f) != lev.size()) { puts("cannot read the plane"); return 1; }
    fclose(f);
    const int TW = 64, TH = 32, tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH;
    double tiles = 0, sum_levels = 0, sum_rounds = 0, sum_runs = 0, sum_rowruns_max = 0;
    for (int j = 0; j < ty; ++j)
        for (int i = 0; i < tx; ++i) {
            // levels present
            bool present[256] = {false};
            for (int y = 0; y < TH; ++y)
                for (int x = 0; x < TW; ++x) {
                    const int gx = i * TW + x, gy = j * TH + y;
                    if (gx < W && gy < H && lev[(size_t)gy * W + gx] != 255) present[lev[(size_t)gy * W + gx]] = true;
                }
            int nlev = 0; double rounds = 0, runs_total = 0;
            for (int t = 0; t < 255; ++t) {
                if (!present[t]) continue;
                ++nlev;
                int mx = 0;
                for (int y = 0; y < TH; ++y) {
                    int runs = 0; bool in = false;
                    for (int x = 0; x < TW; ++x) {
                        const int gx = i * TW + x, gy = j * TH + y;
                        const bool on = gx < W && gy < H && lev[(size_t)gy * W + gx] <= t;
                        if (on && !in) ++runs;
                        in = on;
                    }
                    mx = std::max(mx, runs); runs_total += runs;
                }
                rounds += mx;
            }
            tiles += 1; sum_levels += nlev; sum_rounds += rounds; sum_runs += runs_total;
        }
    const double lv = sum_levels / tiles, rd = sum_rounds / tiles;
    // two tiles per wave: the rounds of a wave are the max over both tiles' rows -- at least one tile's rounds, here priced as the mean of one tile (a lower bound)
    const double per_wave = lv * 20 + rd * 30, per_512 = per_wave / 8;
    printf("%s: %g tiles, levels present per tile %.1f, run rounds per tile %.1f (runs of M_t over all rows and levels: %.0f per tile)\n", argv[1], tiles, lv, rd, sum_runs / tiles);
    printf("  bit-plane form: >= %.0f vector instructions per wave of 4096 pixels = %.0f per 512 pixels (piece kernel, measured: 1320 vector / 2389 all per 512 pixels)\n", per_wave, per_512);
    return 0;
}
