// Host-side model of the GPU entropy decoder's synchronisation pass (dali_amd/csrc/jpeg_huffman.hip: DecodeRange /
// Relax / SyncKernel).  Development tool only: it replays the relaxation on real streams and reports, per workgroup,
// the number of rounds and the length of the critical path (sum over rounds of the longest re-decode), so that slice
// sizes / table shapes / start guesses can be compared without GPU time.
//
//   g++ -O2 -std=c++17 -Iinclude tools/sync_sim.cpp -Ldali_amd/lib -ldali_amd_host -Wl,-rpath,$PWD/dali_amd/lib -o /tmp/sync_sim
//   /tmp/sync_sim DIR [slice_bytes=256] [seg_threads=128] [warm=12] [pair_bits=0]
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <vector>

#include "dali_amd_host.h"

struct State { uint32_t pos, c, z; bool operator==(const State &o) const { return pos == o.pos && c == o.c && z == o.z; } };

static uint32_t MakeEntry(int len, int sym, bool is_dc) {
  int s = sym & 15, r = sym >> 4;
  int adv = is_dc ? 1 : (s ? r + 1 : (r == 15 ? 16 : 64));
  return (uint32_t)((s << 12) | ((len + s) << 7) | adv);
}

struct Image {
  std::vector<uint8_t> clean;
  std::vector<uint16_t> tab[4];  // direct 16-bit tables: slots dc0, dc1, ac0, ac1
  uint32_t dc_mask = 0, ac_mask = 0, bpm = 1;
  uint32_t total_bits = 0;
  // pair tables (pair_bits > 0): index = next pair_bits bits; fields of up to two symbols of the SAME ac table
  struct Pair { uint8_t adv1, z1, adv12, z12, n, zprev, advprev; };
  std::vector<Pair> pair[4];  // per table slot (dc0, dc1, ac0, ac1)
};

static void BuildTable(std::vector<uint16_t> &t, const uint8_t *bits, const uint8_t *vals, bool is_dc) {
  t.assign(65536, (uint16_t)MakeEntry(16, 0, is_dc));
  int code = 0, k = 0;
  for (int l = 1; l <= 16; l++) {
    for (int i = 0; i < bits[l - 1]; i++, k++, code++) {
      uint16_t e = (uint16_t)MakeEntry(l, vals[k], is_dc);
      int first = code << (16 - l), count = 1 << (16 - l);
      for (int j = 0; j < count; j++) t[first + j] = e;
    }
    code <<= 1;
  }
}

static inline uint32_t Peek16(const Image &im, uint32_t pos) {
  size_t b = pos >> 3;
  uint32_t w = ((uint32_t)im.clean[b] << 24) | ((uint32_t)im.clean[b + 1] << 16) | ((uint32_t)im.clean[b + 2] << 8) | im.clean[b + 3];
  return (w << (pos & 7)) >> 16;
}

static int g_pair_bits = 0;
static int g_group = 2;      // symbols per look-up at most
static int g_dc_chain = 0;   // DC entries continue into the AC table of the same class
long g_cls[4] = {0, 0, 0, 0};
long g_pm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
std::vector<long> g_wmax;
std::vector<long> g_nblk;   // histogram: block starts per slice (final decode)

// mirrors DecodeRange: symbols that START in [st.pos, end)
// `prev` (sorted block starts (pos << 8 | c) of the lane's previous decode) makes the decode stop where it meets that
// trajectory at a block start: *met = index into prev.  `rec` receives this decode's block starts.
static int Decode(const Image &im, State &st, uint32_t end, int &nsym, int &steps, const std::vector<uint64_t> *prev = nullptr,
                  std::vector<uint64_t> *rec = nullptr, long *met = nullptr) {
  int nblk = 0;
  nsym = 0; steps = 0;
  uint32_t pos = st.pos, c = st.c, z = st.z;
  if (met) *met = -1;
  while ((int)(end - pos) > 0) {
    const bool is_dc = z == 0;
    if (is_dc) {
      const uint64_t key = ((uint64_t)pos << 8) | c;
      if (prev) {
        auto it = std::lower_bound(prev->begin(), prev->end(), key);
        if (it != prev->end() && *it == key) { *met = it - prev->begin(); break; }
      }
      if (rec) rec->push_back(key);
    }
    const uint32_t slot = (((is_dc ? im.dc_mask : im.ac_mask) >> c) & 1u) + (is_dc ? 0u : 2u);
    uint32_t peek = Peek16(im, pos);
    uint32_t used, zinc;
    int n = 1;
    if (g_pair_bits && (!is_dc || g_dc_chain)) {
      const Image::Pair &p = im.pair[slot][peek >> (16 - g_pair_bits)];
      if (p.n >= 2 && (int)p.advprev < (int)(end - pos) && z + p.zprev < 64) { used = p.adv12; zinc = p.z12; n = p.n; }
      else if (p.n >= 1) { used = p.adv1; zinc = p.z1; }
      else { uint32_t e = im.tab[slot][peek]; used = (e >> 7) & 31; zinc = e & 127; }
    } else {
      uint32_t e = im.tab[slot][peek];
      used = (e >> 7) & 31; zinc = e & 127;
    }
    pos += used; z += zinc; nsym += n; steps++;
    if (z >= 64) { z = 0; c = c + 1 == im.bpm ? 0 : c + 1; nblk++; }
  }
  st.pos = pos; st.c = c; st.z = z;
  return nblk;
}

static bool Load(const std::string &path, Image &im) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  std::vector<uint8_t> data;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
  fclose(f);
  daliamdJpegInfo info;
  daliamdJpegScan sc;
  if (daliamdJpegParse(data.data(), data.size(), &info) != 0) return false;
  if (daliamdJpegAnalyzeScan(data.data(), data.size(), &info, &sc) != 0 || !sc.eligible) return false;
  const uint8_t *p = data.data() + sc.ecs_offset;
  for (int64_t i = 0; i < sc.ecs_length; i++) {
    im.clean.push_back(p[i]);
    if (p[i] == 0xFF && i + 1 < sc.ecs_length && p[i + 1] == 0) i++;
  }
  im.total_bits = (uint32_t)im.clean.size() * 8;
  im.clean.resize(im.clean.size() + 64, 0);
  im.bpm = sc.blocks_per_mcu;
  for (int k = 0; k < sc.blocks_per_mcu; k++) {
    int comp = sc.comp_of_block[k];
    im.dc_mask |= (uint32_t)(sc.dc_sel[comp] & 1) << k;
    im.ac_mask |= (uint32_t)(sc.ac_sel[comp] & 1) << k;
  }
  for (int t = 0; t < 2; t++) {
    BuildTable(im.tab[t], sc.dc_bits[t], sc.dc_vals[t], true);
    BuildTable(im.tab[2 + t], sc.ac_bits[t], sc.ac_vals[t], false);
  }
  if (g_pair_bits) {
    const int B = g_pair_bits;
    for (int t = 0; t < 4; t++) {
      im.pair[t].assign(1u << B, Image::Pair{0, 0, 0, 0, 0, 0, 0});
      if (t < 2 && !g_dc_chain) continue;
      const int ac = t < 2 ? 2 + t : t;  // table the block continues with (same class)
      for (uint32_t w = 0; w < (1u << B); w++) {
        Image::Pair &q = im.pair[t][w];
        uint32_t e1 = im.tab[t][(w << (16 - B)) & 0xFFFF];
        int u1 = (e1 >> 7) & 31, z1 = e1 & 127, s1 = (e1 >> 12) & 15;
        if (u1 - s1 > B) continue;  // the CODE must be inside the index (the magnitude bits need not be)
        q.adv1 = u1; q.z1 = z1; q.n = 1;
        int used = u1, zz = z1, cnt = 1, zprev = 0, advprev = 0;
        while (cnt < g_group && zz < 64 && used < B) {
          uint32_t rest = (w << used) & ((1u << B) - 1);
          uint32_t e2 = im.tab[ac][(rest << (16 - B)) & 0xFFFF];
          int u2 = (e2 >> 7) & 31, z2 = e2 & 127, s2 = (e2 >> 12) & 15;
          if (used + (u2 - s2) > B) break;  // next code not fully determined by the index bits
          zprev = zz; advprev = used;
          used += u2; zz += z2; cnt++;
          if (z2 >= 64) break;
        }
        if (cnt >= 2 && used < 32 && zz < 128) { q.adv12 = used; q.z12 = zz; q.n = cnt; q.zprev = zprev; q.advprev = advprev; }
      }
    }
  }
  return true;
}

int main(int argc, char **argv) {
  if (argc < 2) return 1;
  const int slice = argc > 2 ? atoi(argv[2]) : 256, T = argc > 3 ? atoi(argv[3]) : 128, warm = argc > 4 ? atoi(argv[4]) : 12;
  g_pair_bits = argc > 5 ? atoi(argv[5]) : 0;
  const int r0_bytes = argc > 6 ? atoi(argv[6]) : 0;
  g_group = argc > 7 ? atoi(argv[7]) : 2;
  g_dc_chain = argc > 8 ? atoi(argv[8]) : 0;
  const int overlap = argc > 9 ? atoi(argv[9]) : 0;  // > 0: round 0 starts this many bytes BEFORE the slice (private warm-up)  // > 0: round 0 decodes only the last r0_bytes of each slice
  const int seg_lanes = T - warm;
  std::vector<std::string> files;
  DIR *dp = opendir(argv[1]);
  while (dirent *e = readdir(dp)) if (strstr(e->d_name, ".jpg")) files.push_back(std::string(argv[1]) + "/" + e->d_name);
  closedir(dp);
  std::sort(files.begin(), files.end());
  std::vector<long> wg_path, wg_rounds;
  long total_steps = 0, total_syms = 0, ideal_syms = 0, busy_wave_steps = 0, compact_wave_steps = 0;
  for (auto &fn : files) {
    Image im;
    if (!Load(fn, im)) { fprintf(stderr, "skip %s\n", fn.c_str()); continue; }
    if (getenv("SIM_PHASEMAP")) {
      // Model of the phase-map scheme (HISTORY.md section 9): round A decodes every slice from its first bit with z = 0 for
      // EVERY block index inside the MCU; round B decodes slice k from every distinct state slice k - 1's candidates
      // reached.  Slice k is settled after the two rounds iff the TRUE state at its start is among those states; the
      // others need one more decode each, one after the other along a run of unsettled slices.
      extern long g_pm[8];
      const long n = (im.total_bits / 8 + slice - 1) / slice;
      std::vector<State> truth(n + 1);
      State t{0, 0, 0};
      truth[0] = t;
      for (long k = 0; k < n; k++) {
        const uint32_t end = (uint32_t)std::min<unsigned long long>((unsigned long long)(k + 1) * slice * 8ull, im.total_bits);
        int ns = 0, st = 0;
        if (t.pos < end) Decode(im, t, end, ns, st);
        truth[k + 1] = t;
        g_pm[5] += st;   // steps of the plain sequential decode
      }
      std::vector<std::vector<State>> exits(n);
      for (long k = 0; k < n; k++) {
        const uint32_t begin = (uint32_t)((unsigned long long)k * slice * 8ull);
        const uint32_t end = (uint32_t)std::min<unsigned long long>((unsigned long long)(k + 1) * slice * 8ull, im.total_bits);
        for (uint32_t c = 0; c < im.bpm; c++) {
          State st{begin, c, 0};
          int ns = 0, steps = 0;
          Decode(im, st, end, ns, steps);
          g_pm[0] += steps;   // round A
          bool have = false;
          for (auto &e : exits[k]) have = have || e == st;
          if (!have) exits[k].push_back(st);
        }
        g_pm[1] += (long)exits[k].size();
      }
      long run = 0;
      for (long k = 1; k < n; k++) {
        const uint32_t end = (uint32_t)std::min<unsigned long long>((unsigned long long)(k + 1) * slice * 8ull, im.total_bits);
        bool settled = false;
        for (auto &e : exits[k - 1]) {
          State st = e;
          int ns = 0, steps = 0;
          if (st.pos < end) Decode(im, st, end, ns, steps);
          g_pm[2] += steps;   // round B
          settled = settled || e == truth[k];
        }
        g_pm[3]++;
        if (!settled) { g_pm[4]++; run++; g_pm[6] = std::max(g_pm[6], run); } else run = 0;
      }
      g_pm[7] += n;
      continue;
    }
    const long nslices = (im.total_bits / 8 + slice - 1) / slice;
    const long nseg = std::max<long>(1, (nslices + seg_lanes - 1) / seg_lanes);
    for (long seg = 0; seg < nseg; seg++) {
      struct L { uint32_t begin, end; bool active; State in, out; bool has_in; int nsym, steps; std::vector<uint64_t> traj;
                 std::vector<std::pair<State, State>> memo; };   // SIM_SPEC: in -> out of speculative decodes
      std::vector<L> ln(T);
      std::vector<State> state(T);
      for (int t = 0; t < T; t++) {
        long si = seg * seg_lanes + t - warm;
        L &l = ln[t];
        l.has_in = false;
        if (si < 0) { l.begin = l.end = 0; l.active = false; }
        else {
          unsigned long long b = (unsigned long long)si * slice * 8ull;
          l.begin = (uint32_t)std::min<unsigned long long>(b, im.total_bits);
          l.end = (uint32_t)std::min<unsigned long long>(b + slice * 8ull, im.total_bits);
          l.active = l.begin < im.total_bits;
        }
        state[t] = State{l.begin, 0, 0};
        l.nsym = l.steps = 0;
      }
      long path = 0, rounds = 0;
      std::string trace;
      for (int round = 0; round <= T; round++) {
        long wave_max[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        std::vector<long> round_steps;   // of the lanes that decode in this round, in lane order (compaction model)
        bool any = false;
        for (int t = 0; t < T; t++) {
          L &l = ln[t];
          static const bool spec = getenv("SIM_SPEC") != nullptr;
          auto speculate = [&](int tt) {
            // idle lanes decode slice tt from its present input with every other block index: if its predecessor's
            // result changes in the block index only, this lane's result is there already
            if (!(tt < T && ln[tt].active && ln[tt].has_in)) return;
            L &nx = ln[tt];
            for (uint32_t c = 0; c < im.bpm; c++) {
              State cand = nx.in;
              if (c == cand.c) continue;
              cand.c = c;
              bool have = false;
              for (auto &m : nx.memo) have = have || m.first == cand;
              if (have) continue;
              State st2 = cand;
              int ns = 0, stp = 0;
              if (st2.pos < nx.end) Decode(im, st2, nx.end, ns, stp);
              nx.memo.push_back({cand, st2});
              wave_max[tt / 64] = std::max<long>(wave_max[tt / 64], stp);
              total_steps += stp;
            }
          };
          if (spec && l.active && l.has_in && !(l.in == state[t])) {   // a result decoded speculatively in an earlier round?
            bool hit = false;
            for (auto &m : l.memo)
              if (m.first == state[t]) { l.in = state[t]; l.out = m.second; hit = true; break; }
            if (hit) {                      // free: the state travels on within this round
              if (t + 1 < T && !(state[t + 1] == l.out)) state[t + 1] = l.out;
              if (getenv("SIM_SPEC")[0] == '2') speculate(t + 2);   // keep one lane ahead of the travelling state
              any = true;
              continue;
            }
          }
          if (l.active && !(l.has_in && l.in == state[t])) {
            if (spec && round >= 2) speculate(t + 1);
            l.in = state[t]; l.has_in = true;
            const State old_out = l.out; const bool had = l.has_in;
            const State old_in = l.in;
            State st = l.in;
            if (round == 0 && r0_bytes > 0 && t > 0 && l.end - l.begin > (uint32_t)r0_bytes * 8) {
              st.pos = l.end - r0_bytes * 8;
              l.has_in = false;  // a partial decode: never accepted as the final one
            }
            l.nsym = l.steps = 0;
            int warm_steps = 0;
            if (round == 0 && overlap > 0 && t > 0 && l.begin >= (uint32_t)overlap * 8) {
              State w{l.begin - (uint32_t)overlap * 8, 0, 0};
              int ns = 0;
              Decode(im, w, l.begin, ns, warm_steps);
              st = w;              // whatever state the private warm-up reached at the start of the slice
              l.has_in = false;    // a guess, never accepted as the final decode
            }
            static const bool early = getenv("SIM_EARLY") != nullptr;
            if (st.pos < l.end) {
              if (early && round >= 1 && !l.traj.empty()) {
                std::vector<uint64_t> rec;
                long met = -1;
                Decode(im, st, l.end, l.nsym, l.steps, &l.traj, &rec, &met);
                if (met >= 0) {  // the rest is the previous trajectory: keep its tail and its out state
                  rec.insert(rec.end(), l.traj.begin() + met, l.traj.end());
                  st = old_out;
                }
                l.traj.swap(rec);
              } else {
                l.traj.clear();
                Decode(im, st, l.end, l.nsym, l.steps, nullptr, &l.traj, nullptr);
              }
            }
            l.steps += warm_steps;
            l.out = st;
            if (getenv("SIM_LANES") && round >= 2 && fn.find(getenv("SIM_LANES")) != std::string::npos)
              printf("    %s seg %ld round %d lane %d in (%u,%u,%u) was (%u,%u,%u) -> out (%u,%u,%u) was (%u,%u,%u) steps %d\n", fn.c_str() + fn.size() - 8, seg, round, t, l.in.pos, l.in.c, l.in.z, old_in.pos, old_in.c, old_in.z, st.pos, st.c, st.z, old_out.pos, old_out.c, old_out.z, l.steps);
            if (had && round >= 2) {
              extern long g_cls[4];
              if (st == old_out) g_cls[0]++; else if (st.pos == old_out.pos && st.z == old_out.z) g_cls[1]++; else g_cls[2]++;
            }
            wave_max[t / 64] = std::max<long>(wave_max[t / 64], l.steps);
            round_steps.push_back(l.steps);
            total_steps += l.steps; total_syms += l.nsym;
            any = true;
          }
        }
        long m = 0;
        for (int w = 0; w < T / 64; w++) { m = std::max(m, wave_max[w]); busy_wave_steps += wave_max[w]; }
        // the same round with the decoding lanes packed into the first waves (work list): a wave lasts as long as its
        // longest lane, waves without work issue nothing
        for (size_t i = 0; i < round_steps.size(); i += 64) {
          long wm = 0;
          for (size_t j = i; j < std::min(round_steps.size(), i + 64); j++) wm = std::max(wm, round_steps[j]);
          compact_wave_steps += wm;
        }
        path += m;
        if (any) { rounds++; trace += " " + std::to_string(m); }
        bool changed = false;
        for (int t = 0; t + 1 < T; t++)
          if (ln[t].active && !(state[t + 1] == ln[t].out)) { state[t + 1] = ln[t].out; changed = true; }
        if (!changed) break;
      }
      { extern std::vector<long> g_wmax; long m = 0; for (int t = warm; t < T; t++) m = std::max<long>(m, ln[t].nsym); g_wmax.push_back(m); }
      for (int t = warm; t < T; t++) ideal_syms += ln[t].nsym;
      { extern std::vector<long> g_nblk; for (int t = warm; t < T; t++) if (ln[t].active) { size_t n = ln[t].traj.size(); if (g_nblk.size() <= n) g_nblk.resize(n + 1); g_nblk[n]++; } }
      if (getenv("SIM_TRACE") && path > atol(getenv("SIM_TRACE")))
        printf("  %s seg %ld/%ld path %ld rounds:%s\n", fn.c_str() + fn.size() - 8, seg, nseg, path, trace.c_str());
      wg_path.push_back(path);
      wg_rounds.push_back(rounds);
    }
  }
  if (getenv("SIM_PHASEMAP")) {
    printf("phase map, slice %d: %ld slices; distinct exit states per slice %.2f; lane-steps round A %ld + round B %ld = %.2f x the "
           "sequential decode (%ld); slices not settled by the two rounds %ld of %ld (%.3f), longest run of them %ld\n",
           slice, g_pm[7], (double)g_pm[1] / g_pm[7], g_pm[0], g_pm[2], (double)(g_pm[0] + g_pm[2]) / g_pm[5], g_pm[5], g_pm[4], g_pm[3],
           (double)g_pm[4] / g_pm[3], g_pm[6]);
    return 0;
  }
  std::sort(wg_path.begin(), wg_path.end());
  std::sort(wg_rounds.begin(), wg_rounds.end());
  auto pct = [&](std::vector<long> &v, double p) { return v[(size_t)std::min<double>(v.size() - 1, p * v.size())]; };
  double mean_path = 0; for (long p : wg_path) mean_path += p; mean_path /= wg_path.size();
  printf("slice %d  threads %d  warm %d  pair_bits %d\n", slice, T, warm, g_pair_bits);
  printf("workgroups %zu   symbols (final states) %ld   decoded symbol-executions %ld (x%.2f)  lane-steps %ld\n", wg_path.size(),
         ideal_syms, total_syms, (double)total_syms / ideal_syms, total_steps);
  printf("rounds: p50 %ld p90 %ld p99 %ld max %ld\n", pct(wg_rounds, .5), pct(wg_rounds, .9), pct(wg_rounds, .99), wg_rounds.back());
  printf("critical path (steps): mean %.0f p50 %ld p90 %ld p99 %ld max %ld   wave-steps total %ld\n", mean_path, pct(wg_path, .5),
         pct(wg_path, .9), pct(wg_path, .99), wg_path.back(), busy_wave_steps);
  printf("wave-steps with the decoding lanes of a round packed into the first waves: %ld (%.3f of the above)\n", compact_wave_steps,
         (double)compact_wave_steps / busy_wave_steps);
  std::sort(g_wmax.begin(), g_wmax.end());
  printf("write pass: longest lane (symbols) per workgroup: p50 %ld p90 %ld p99 %ld max %ld\n", pct(g_wmax, .5), pct(g_wmax, .9), pct(g_wmax, .99), g_wmax.back());
  { long tot = 0, acc = 0; for (long v : g_nblk) tot += v;
    printf("block starts per slice: ");
    for (size_t n = 0; n < g_nblk.size(); n++) { acc += g_nblk[n]; if (n == 17 || n == 21 || n == 25 || n == 29 || n == 33 || n == 41 || n == 65) printf(" <=%zu: %.4f", n, (double)acc / tot); }
    printf("  max %zu\n", g_nblk.size() - 1); }
  printf("re-decodes in rounds >= 2: same out %ld, same (pos,z) other c %ld, other pos %ld\n", g_cls[0], g_cls[1], g_cls[2]);
  return 0;
}
