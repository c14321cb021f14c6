// TEST INFRASTRUCTURE: see mini_runtime.h
#include "google/protobuf/mini_runtime.h"
