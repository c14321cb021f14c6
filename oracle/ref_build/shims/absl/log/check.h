// TEST INFRASTRUCTURE (oracle/ref_build): CHECK family; a failed check throws dvref_log::Fatal.
#ifndef DVREF_ABSL_CHECK_H_
#define DVREF_ABSL_CHECK_H_
#include "absl/log/log.h"
#define CHECK(cond) (cond) ? (void)0 : ::dvref_log::Voidify() & ::dvref_log::Message(__FILE__, __LINE__, 3) << "Check failed: " #cond " "
#define DVREF_CHECK_OP(a, op, b) ((a)op(b)) ? (void)0 : ::dvref_log::Voidify() & ::dvref_log::Message(__FILE__, __LINE__, 3) << "Check failed: " #a " " #op " " #b " "
#define CHECK_EQ(a, b) DVREF_CHECK_OP(a, ==, b)
#define CHECK_NE(a, b) DVREF_CHECK_OP(a, !=, b)
#define CHECK_LT(a, b) DVREF_CHECK_OP(a, <, b)
#define CHECK_LE(a, b) DVREF_CHECK_OP(a, <=, b)
#define CHECK_GT(a, b) DVREF_CHECK_OP(a, >, b)
#define CHECK_GE(a, b) DVREF_CHECK_OP(a, >=, b)
#define QCHECK(cond) CHECK(cond)
#define QCHECK_EQ(a, b) CHECK_EQ(a, b)
#define DCHECK(cond) CHECK(cond)
#define DCHECK_EQ(a, b) CHECK_EQ(a, b)
#define DCHECK_NE(a, b) CHECK_NE(a, b)
#define DCHECK_LT(a, b) CHECK_LT(a, b)
#define DCHECK_LE(a, b) CHECK_LE(a, b)
#define DCHECK_GT(a, b) CHECK_GT(a, b)
#define DCHECK_GE(a, b) CHECK_GE(a, b)
#define CHECK_OK(expr) CHECK((expr).ok())
#endif
