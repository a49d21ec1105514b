// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. include/core/parameters.hpp only names nlohmann::json in two member declarations.
#pragma once
namespace nlohmann { class json; }
