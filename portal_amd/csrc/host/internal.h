// internal.h -- shared between the translation units of libportal_amd.so.
#pragma once
#include <string>

namespace ptl {
void set_last_error(const std::string& msg);
extern thread_local std::string g_last_error;
}  // namespace ptl
