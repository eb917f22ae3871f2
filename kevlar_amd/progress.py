"""Progress lines with decaying frequency (the reference's kevlar/progress.py:13-42)."""
import kevlar_amd


class ProgressIndicator(object):
    def __init__(self, message, interval=10, breaks=(100, 1000, 10000), usetimer=False):
        self.message = message
        self.counter = 0
        self.interval = interval
        self.nextupdate = interval
        self.breaks = breaks
        self.timer = None
        if usetimer:
            self.timer = kevlar_amd.Timer()
            self.timer.start()

    def update(self, n=1):
        """Advance by n items (n > 1 lets batch drivers keep the same log cadence)."""
        for _ in range(n):
            if self.counter in self.breaks:
                self.interval = self.counter
            if self.counter >= self.nextupdate:
                self.nextupdate += self.interval
                message = self.message.format(counter=self.counter)
                if self.timer:
                    message += ' ({:.2f} seconds elapsed)'.format(self.timer.probe())
                kevlar_amd.plog(message)
            self.counter += 1
