// _elfgames_go_inference / _elfgames_go: the reference's two game modules (elfgames/go/inference/pybind_module.cc,
// elfgames/go/train/pybind_module.cc) as thin re-exports of the classes _elf registers (pybind_elf.cc), so that every C++ type
// lives in one shared object.  Compiled twice: -DELF_GO_MODULE_INFERENCE / -DELF_GO_MODULE_TRAIN.
#include <pybind11/pybind11.h>

namespace py = pybind11;

static void reexport(py::module_& m, bool inference) {
  py::module_ go = py::module_::import("_elf").attr("_go");
  for (const char* name : {"ContextOptions", "GameOptions", "GoGameSelfPlay"}) m.attr(name) = go.attr(name);
  m.attr("GameContext") = go.attr(inference ? "GameContextInference" : "GameContextTrain");
  if (!inference)
    for (const char* name : {"Client", "Server", "GameStats", "WinRateStats"}) m.attr(name) = go.attr(name);
}

#if defined(ELF_GO_MODULE_INFERENCE)
PYBIND11_MODULE(_elfgames_go_inference, m) { reexport(m, true); }
#elif defined(ELF_GO_MODULE_TRAIN)
PYBIND11_MODULE(_elfgames_go, m) { reexport(m, false); }
#else
#error "define ELF_GO_MODULE_INFERENCE or ELF_GO_MODULE_TRAIN"
#endif
