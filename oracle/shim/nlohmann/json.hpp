#pragma once
#include <json.hpp>
