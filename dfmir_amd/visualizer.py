"""Console / loss_log.txt reporting -- the step right AFTER the hot path (SURVEY.md section 8 row N4;
reference util/visualizer.py:46-93,226-242).  visdom / HTML pages are out of scope."""
import os
import time


class Visualizer(object):
    def __init__(self, opt):
        self.opt = opt
        self.name = opt.name
        d = os.path.join(opt.checkpoints_dir, opt.name)
        os.makedirs(d, exist_ok=True)
        self.log_name = os.path.join(d, 'loss_log.txt')
        with open(self.log_name, "a") as f:
            f.write('================ Training Loss (%s) ================\n' % time.strftime("%c"))

    def reset(self):
        pass

    def print_current_losses(self, epoch, iters, losses, t_comp, t_data):
        message = '(epoch: %d, iters: %d, time: %.3f, data: %.3f) ' % (epoch, iters, t_comp, t_data)
        for k, v in losses.items():
            message += '%s: %.3f ' % (k, v)
        print(message)
        with open(self.log_name, "a") as f:
            f.write('%s\n' % message)
        return message
