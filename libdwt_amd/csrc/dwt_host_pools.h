// dwt_host_pools.h -- the library's two HOST thread pools, plain C++ (no HIP): the row pool of the host-pointer
// calls' repacking (dwt_host_xfer.hip) and the per-slot worker threads of the multi-GPU entries (dwt_multi.hip).
// In a header of their own so that tests/san/tsan_pools.cpp can run them under ThreadSanitizer with host-only jobs
// (`make -C tests/san tsan`; GPU code cannot run under sanitizers on this pool).
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

namespace dwtb {

// A small persistent pool for the host-side row repacking (starting 16 threads per call cost more
// than the repacking of a 1080p frame).  Workers sleep on a condition variable between jobs; a job
// is a range of row chunks handed out under the mutex; the caller works too.  One job at a time
// (calls from several host threads take turns).  The pool is created on first use and never
// destroyed (no static-destruction order to get wrong); a forked child builds its own.
class RowPool {
public:
	static RowPool &get()
	{
		static RowPool *p = nullptr;
		static std::mutex mk;
		std::lock_guard<std::mutex> lk(mk);
		if (!p || p->pid_ != getpid())
			p = new RowPool();
		return *p;
	}
	template <class F>
	void run(int rows, int chunk, F f)
	{
		std::lock_guard<std::mutex> turn(turn_);
		std::function<void(int, int)> fn = f;
		{
			std::lock_guard<std::mutex> lk(m_);
			job_ = &fn; rows_ = rows; chunk_ = chunk; next_ = 0; active_ = 0; gen_++;
		}
		cv_job_.notify_all();
		work();
		std::unique_lock<std::mutex> lk(m_);
		cv_done_.wait(lk, [&] { return next_ >= rows_ && active_ == 0; });
		job_ = nullptr;
	}
	int workers() const { return (int)th_.size() + 1; }

private:
	RowPool() : pid_(getpid())
	{
		unsigned n = std::thread::hardware_concurrency();
		n = n > 16 ? 16 : n;
		for (unsigned i = 1; i < n; i++)
			th_.emplace_back([this] { loop(); });
		for (auto &t : th_)
			t.detach();
	}
	void loop()
	{
		unsigned long seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> lk(m_);
				cv_job_.wait(lk, [&] { return gen_ != seen; });
				seen = gen_;
			}
			work();
		}
	}
	void work()
	{
		for (;;) {
			int a, b;
			const std::function<void(int, int)> *fn;
			{
				std::lock_guard<std::mutex> lk(m_);
				if (!job_ || next_ >= rows_)
					break;
				a = next_; b = a + chunk_ < rows_ ? a + chunk_ : rows_;
				next_ = b; active_++; fn = job_;
			}
			(*fn)(a, b);
			{
				std::lock_guard<std::mutex> lk(m_);
				active_--;
			}
			cv_done_.notify_all();
		}
		cv_done_.notify_all();
	}
	pid_t pid_;
	std::vector<std::thread> th_;
	std::mutex m_, turn_;
	std::condition_variable cv_job_, cv_done_;
	const std::function<void(int, int)> *job_ = nullptr;
	int rows_ = 0, chunk_ = 1, next_ = 0, active_ = 0;
	unsigned long gen_ = 0;
};

// A host thread that lives as long as the process and runs one job at a time: the thread-local state of whatever it
// runs (the backend's per-thread context) persists between the calls.  `last_error` reads the failing job's message
// ON THE WORKER THREAD (the message is thread-local there).
class SlotThread {
public:
	explicit SlotThread(const char *(*last_error)(void)) : last_error_(last_error), th_([this] { loop(); }) { th_.detach(); }
	void submit(std::function<int()> job)
	{
		std::lock_guard<std::mutex> lk(m_);
		job_ = std::move(job);
		busy_ = true;
		rc_ = 0;
		cv_.notify_all();
	}
	int wait(std::string &err)
	{
		std::unique_lock<std::mutex> lk(m_);
		cv_.wait(lk, [&] { return !busy_; });
		err = err_;
		return rc_;
	}

private:
	void loop()
	{
		for (;;) {
			std::function<int()> job;
			{
				std::unique_lock<std::mutex> lk(m_);
				cv_.wait(lk, [&] { return busy_ && job_; });
				job = std::move(job_);
				job_ = nullptr;
			}
			const int rc = job();
			{
				std::lock_guard<std::mutex> lk(m_);
				rc_ = rc;
				err_ = rc ? last_error_() : "";
				busy_ = false;
			}
			cv_.notify_all();
		}
	}
	const char *(*last_error_)(void);
	std::mutex m_;
	std::condition_variable cv_;
	std::function<int()> job_;
	bool busy_ = false;
	int rc_ = 0;
	std::string err_;
	std::thread th_;
};

} // namespace dwtb
