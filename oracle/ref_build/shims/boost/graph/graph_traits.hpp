#include "boost/graph/adjacency_list.hpp"
