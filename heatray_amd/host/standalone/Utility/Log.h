// Standalone builds only: the slice of the application's logging interface the drop-in layer and its callers use
// (/root/reference/Source/Utility/Log.h:18-61: util::Log with an installable instance, LOG_INFO / LOG_WARNING / LOG_ERROR).
// In the Heatray tree the application's own Utility/Log.h is found instead (this directory is not on its include path) and
// messages go to whatever logger the viewer installed (ImGuiLog).  Here a message goes to stderr unless a logger was installed.
#pragma once

#include <cstdio>
#include <memory>
#include <string>
#include <string_view>

namespace util {

class Log
{
public:
    enum class Type { kInfo, kWarning, kError, kCount };

    virtual ~Log() = default;
    static std::shared_ptr<Log> instance() { return slot() ? slot() : fallback(); }

    template <class... Args> void log(Type type, const std::string_view format, Args&&... args)
    {
        const std::string fmt(format);
        std::string text;
        if constexpr (sizeof...(Args) == 0) {
            text = fmt;
        } else {
            const int n = std::snprintf(nullptr, 0, fmt.c_str(), args...);
            text.resize(n > 0 ? (size_t)n : 0);
            if (n > 0) std::snprintf(text.data(), (size_t)n + 1, fmt.c_str(), args...);
        }
        addNewItem(text, type);
    }

protected:
    Log() = default;
    static void setInstance(std::shared_ptr<Log> instance) { slot() = std::move(instance); }
    virtual void addNewItem(const std::string_view item, const Type type) = 0;

private:
    static std::shared_ptr<Log>& slot()
    {
        static std::shared_ptr<Log> installed;
        return installed;
    }
    struct Stderr;
    static std::shared_ptr<Log> fallback();
};

struct Log::Stderr final : Log {
    void addNewItem(const std::string_view item, const Type type) override
    {
        static const char* const tag[] = {"info", "warning", "error", "?"};
        std::fprintf(stderr, "[%s] %.*s\n", tag[(int)type], (int)item.size(), item.data());
    }
};
inline std::shared_ptr<Log> Log::fallback()
{
    static std::shared_ptr<Log> s = std::make_shared<Stderr>();
    return s;
}

} // namespace util

#define LOG_INFO(format, ...) util::Log::instance()->log(util::Log::Type::kInfo, format, ##__VA_ARGS__)
#define LOG_WARNING(format, ...) util::Log::instance()->log(util::Log::Type::kWarning, format, ##__VA_ARGS__)
#define LOG_ERROR(format, ...) util::Log::instance()->log(util::Log::Type::kError, format, ##__VA_ARGS__)
