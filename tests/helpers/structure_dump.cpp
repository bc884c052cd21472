// structure_dump.cpp — test helper: runs hs::build_visual_structure (hyperslam_amd/csrc/host_structure.hpp, the host code in front of every
// optimize()) on the tables of a binary input file and dumps every array it produces; tests/test_host_structure.py compares them with an
// independent numpy restatement of the ordering rules. Host-only program (compiled with hipcc because the header shares constants with the kernels).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "host_structure.hpp"

static std::vector<int> read_ints(FILE* f, int n) {
  std::vector<int> v(n);
  if (n && fread(v.data(), sizeof(int), n, f) != size_t(n)) exit(2);
  return v;
}
static std::vector<double> read_doubles(FILE* f, int n) {
  std::vector<double> v(n);
  if (n && fread(v.data(), sizeof(double), n, f) != size_t(n)) exit(2);
  return v;
}
static void dump(FILE* f, const char* name, const std::vector<int>& v) {
  fprintf(f, "%s %zu", name, v.size());
  for (int x : v) fprintf(f, " %d", x);
  fprintf(f, "\n");
}

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  FILE* in = fopen(argv[1], "rb");
  if (!in) return 1;
  int hdr[5];  // k, n_cp, n_lm, n_px, n_br
  double t[2];  // t0, dt
  if (fread(hdr, sizeof(int), 5, in) != 5 || fread(t, sizeof(double), 2, in) != 2) return 2;
  const std::vector<double> px_stamp = read_doubles(in, hdr[3]), br_stamp = read_doubles(in, hdr[4]);
  const std::vector<int> px_lm = read_ints(in, hdr[3]), br_lm = read_ints(in, hdr[4]);
  fclose(in);
  hs::VisualInput vi = {hdr[0], hdr[1], hdr[2], t[0], t[1], hdr[3], hdr[4], px_stamp.data(), br_stamp.data(), px_lm.data(), br_lm.data()};
  hs::VisualStructure vs;
  std::string err;
  FILE* out = fopen(argv[2], "w");
  for (int rep = 0; rep < 2; ++rep) {  // twice into the same object: the work arrays are reused from call to call
    if (!hs::build_visual_structure(vi, &vs, &err)) {
      fprintf(out, "error %s\n", err.c_str());
      fclose(out);
      return 0;
    }
  }
  dump(out, "table_type", vs.table_type), dump(out, "table_idx", vs.table_idx), dump(out, "lm_dev", vs.lm_dev), dump(out, "first", vs.first);
  dump(out, "pos", vs.pos), dump(out, "seg_ptr", vs.seg_ptr), dump(out, "dev_of_table", vs.dev_of_table), dump(out, "table_of_dev", vs.table_of_dev);
  dump(out, "lm_ptr", vs.lm_ptr), dump(out, "lm_cfirst", vs.lm_cfirst), dump(out, "lm_ncp", vs.lm_ncp), dump(out, "lm_yoff", vs.lm_yoff);
  dump(out, "cf_ptr", vs.cf_ptr);
  fprintf(out, "bw 1 %d\ny_total 1 %d\n", vs.bw, vs.y_total);
  // chunks of the fused build (build_chunks) and their dispatch order on machines of 8 and 256 compute units (order_chunks_for_dispatch)
  const int R = argc > 3 ? atoi(argv[3]) : 138, L = argc > 4 ? atoi(argv[4]) : 14;
  std::vector<int> ch_ptr, gw_ptr, gw_cf, ch_desc;
  if (hs::build_chunks(vs, vi.n_cp, R, L, &ch_ptr, &gw_ptr, &gw_cf, &ch_desc)) {
    const int n = int(ch_ptr.size()) - 1;
    dump(out, "ch_ptr", ch_ptr), dump(out, "gw_ptr", gw_ptr), dump(out, "gw_cf", gw_cf), dump(out, "ch_desc", ch_desc);
    for (int n_cu : {8, 256}) {
      std::vector<int> d = ch_desc;
      hs::order_chunks_for_dispatch(vs, vi.k, n_cu, &d, n);
      dump(out, n_cu == 8 ? "ch_desc_cu8" : "ch_desc_cu256", d);
    }
  } else {
    fprintf(out, "ch_ptr 0\n");
  }
  fclose(out);
  return 0;
}
