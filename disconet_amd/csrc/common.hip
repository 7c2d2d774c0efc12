#include "dn_internal.h"

namespace dn {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace dn

extern "C" int dn_version(void) { return 100; }  // 0.1.0

extern "C" const char* dn_last_error(void) { return dn::err_buf(); }
