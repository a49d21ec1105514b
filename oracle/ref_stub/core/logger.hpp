// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. Stand-in for the reference's include/core/logger.hpp (spdlog + <format>, neither in this image) so that its loader
// sources compile in place: the LOG_* macros expand to nothing (their arguments are not evaluated), and std::format - which libstdc++ 11 does not ship and the
// loaders also use to build exception texts - is a minimal "{}" / "{:spec}" substitution through operator<<.
#pragma once
#include <sstream>
#include <string>
#include <string_view>

namespace std {
    namespace ref_stub_detail {
        inline void emit(std::ostringstream& os, std::string_view& f) {
            os << f;
            f = {};
        }
        template <class T, class... R> inline void emit(std::ostringstream& os, std::string_view& f, const T& v, const R&... rest) {
            const size_t a = f.find('{');
            if (a == std::string_view::npos) {
                os << f;
                f = {};
                return;
            }
            const size_t b = f.find('}', a);
            os << f.substr(0, a) << v;
            f = f.substr(b + 1);
            emit(os, f, rest...);
        }
    } // namespace ref_stub_detail
    template <class... A> inline std::string format(std::string_view f, const A&... a) {
        std::ostringstream os;
        ref_stub_detail::emit(os, f, a...);
        return os.str();
    }
    template <class... A> inline void println(std::string_view, const A&...) {} // <print> (C++23): progress messages only
} // namespace std

#define LOG_TRACE(...)       ((void)0)
#define LOG_DEBUG(...)       ((void)0)
#define LOG_INFO(...)        ((void)0)
#define LOG_WARN(...)        ((void)0)
#define LOG_ERROR(...)       ((void)0)
#define LOG_CRITICAL(...)    ((void)0)
#define LOG_TIMER(name)      ((void)0)
#define LOG_TIMER_TRACE(name) ((void)0)
