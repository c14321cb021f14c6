#include "absl/container/flat_hash_set.h"
