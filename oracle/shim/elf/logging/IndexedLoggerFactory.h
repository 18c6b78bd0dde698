#pragma once
#include <spdlog/spdlog.h>
namespace elf { namespace logging {
inline std::shared_ptr<spdlog::logger> getIndexedLogger(const std::string&, const std::string&) {
  return std::make_shared<spdlog::logger>();
}
}}
