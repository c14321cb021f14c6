#include "absl/random/random.h"
