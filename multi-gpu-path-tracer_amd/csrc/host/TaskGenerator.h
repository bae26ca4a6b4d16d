// TaskGenerator.h -- equal-size task rectangles (Scheduling/TaskGenerator.h:46-80) through
// the C ABI's generators, returned as RenderTask vectors like the reference.
#pragma once

#include <vector>

#include "DevicePathTracer.h"

namespace ptamd {

class TaskGenerator {
public:
    TaskGenerator() = default;
    TaskGenerator(int width, int height) : width_{width}, height_{height} {}

    // tasks of equal width in one row, the last one absorbs the remainder
    std::vector<RenderTask> generateEqualTasks(int task_count, int width, int height)
    {
        std::vector<pt_task> raw((size_t)task_count);
        checkPtErrors(pt_tasks_equal_row(task_count, width, height, raw.data()));
        return convert(raw);
    }

    // taskLayout[i] lists the task ids of screen row i
    std::vector<RenderTask> generateEqualTasks(int taskCount, std::vector<std::vector<int>> &taskLayout, int width, int height)
    {
        size_t maxRow = 1;
        for (auto &row : taskLayout) maxRow = row.size() > maxRow ? row.size() : maxRow;
        std::vector<pt_task> raw((size_t)taskCount);
        checkPtErrors(pt_tasks_equal_layout(taskCount, (int)maxRow, width, height, raw.data()));
        return convert(raw);
    }

    void setRes(int width, int height) { width_ = width; height_ = height; }

private:
    static std::vector<RenderTask> convert(const std::vector<pt_task> &raw)
    {
        std::vector<RenderTask> out;
        out.reserve(raw.size());
        for (const pt_task &t : raw) out.push_back(RenderTask{t.width, t.height, t.offset_x, t.offset_y});
        return out;
    }
    int width_ = 0;
    int height_ = 0;
};

} // namespace ptamd
