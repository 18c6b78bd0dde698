#pragma once
#include <memory>
#include <string>
namespace spdlog {
class logger {
 public:
  template <typename... A> void trace(const A&...) {}
  template <typename... A> void debug(const A&...) {}
  template <typename... A> void info(const A&...) {}
  template <typename... A> void warn(const A&...) {}
  template <typename... A> void error(const A&...) {}
  template <typename... A> void critical(const A&...) {}
};
}
