#include "absl/status/status.h"
