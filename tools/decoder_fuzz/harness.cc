// ASAN/UBSAN harness for the host-only alignment-file decoders: reads every file named on the command line with all
// plane parsing on, prints status.  Not part of the product.
#include <cstdio>
#include <string>
#include "dv_internal.h"
namespace dv {
static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
int fail(int status, const std::string& m) { g_err = m; return status; }
}
extern "C" const char* dv_last_error(void) { return dv::g_err.c_str(); }
int main(int argc, char** argv) {
  for (int i = 1; i < argc; ++i) {
    dv_read_requirements rq{};
    rq.keep_duplicates = rq.keep_failed_vendor_quality_checks = rq.keep_secondary_alignments = rq.keep_supplementary_alignments = rq.keep_improperly_placed = 1;
    rq.parse_base_modifications = rq.parse_flow_tags = 1;
    dv_read_table* t = nullptr;
    std::string p = argv[i];
    int rc = p.size() > 5 && p.substr(p.size() - 5) == ".cram" ? dv_cram_read_region(argv[i], nullptr, 0, 1ll << 40, &rq, nullptr, nullptr, 2, &t)
                                                                 : dv_bam_read_region(argv[i], nullptr, 0, 1ll << 40, &rq, 2, &t);
    printf("%s rc=%d %s\n", argv[i], rc, rc ? dv_last_error() : "");
    if (t) dv_read_table_free(t);
  }
  return 0;
}
