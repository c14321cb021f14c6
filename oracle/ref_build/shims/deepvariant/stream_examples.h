// TEST INFRASTRUCTURE (oracle/ref_build): shadows deepvariant/stream_examples.h (boost::interprocess shared memory,
// the fast_pipeline hand-over this repository replaces with the fused device route).
// ExamplesGenerator only constructs one when MakeExamplesOptions.stream_examples is set, which nothing here does;
// the class exists so that make_examples_native.cc compiles unmodified.
#ifndef DVREF_STREAM_EXAMPLES_SHIM_H_
#define DVREF_STREAM_EXAMPLES_SHIM_H_
#include <memory>
#include <string>
#include <vector>
#include "absl/log/log.h"
#include "absl/strings/string_view.h"
#include "absl/types/span.h"
#include "deepvariant/pileup_image_native.h"
#include "deepvariant/protos/deepvariant.pb.h"
namespace learning {
namespace genomics {
namespace deepvariant {
class StreamExamples {
 public:
  StreamExamples(const MakeExamplesOptions&, const AltAlignedPileup&) {
    LOG(FATAL) << "stream_examples is not part of the oracle/_ref build";
  }
  void StartStreaming() {}
  void EndStreaming(bool) {}
  void SignalShardFinished() {}
  void StreamExample(std::vector<std::vector<std::unique_ptr<ImageRow>>>&,
                     std::vector<std::vector<std::vector<std::unique_ptr<ImageRow>>>>&, const AltAlignedPileup&,
                     absl::Span<const std::string>, absl::string_view, absl::string_view) {}
};
}  // namespace deepvariant
}  // namespace genomics
}  // namespace learning
#endif
