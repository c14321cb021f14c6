#include "absl/container/flat_hash_map.h"
