"""The offline experiment loops of the reference's ``main.py`` on the MI355X path.

Reference: main.py -- flags :24-51, create_model :56-75, train :78-202 (feed -> step, every
``steps_per_checkpoint`` steps: statistics, validation through ``get_next_batch`` + ``merge_TFSummary``
:153-165, optional test pass :173-178, best-checkpoint rule on ``objective_metric`` :180-194, stop on
an infinite loss or past ``max_train_iteration`` :195-202), test :206-262 (scores of the whole test
set, merged summary, TREC rank lists through ``output_ranklist``).

Same settings JSON, same flag names, same printed lines.  The session is gone (``None`` is passed where
the reference passes it), TensorBoard writers are replaced by the returned history, and a checkpoint is
the ``.pt`` file ``Saver`` writes plus a ``checkpoint`` index file holding its path (what
``tf.train.get_checkpoint_state`` reads in the reference).

    python -m ultra_amd.main --data_dir ./tests/golden/toy_ultra/ --setting_file settings.json \\
        --model_dir /tmp/model --output_dir /tmp/out --max_train_iteration 100
"""
import argparse
import json
import os
import sys
import time

from .utils import find_class, merge_TFSummary, output_ranklist, read_data


class Flags(object):
    """The reference's tf.app.flags (main.py:24-51) as a plain object."""

    def __init__(self, **kw):
        self.data_dir = './tests/data/'
        self.train_data_prefix = 'train'
        self.valid_data_prefix = 'valid'
        self.test_data_prefix = 'test'
        self.model_dir = './tests/tmp_model/'
        self.output_dir = './tests/tmp_output/'
        self.setting_file = './example/offline_setting/dla_exp_settings.json'
        self.batch_size = 256
        self.max_list_cutoff = 0
        self.selection_bias_cutoff = 10
        self.max_train_iteration = 10000
        self.start_saving_iteration = 0
        self.steps_per_checkpoint = 50
        self.test_while_train = False
        self.test_only = False
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError('unknown flag %r' % k)
            setattr(self, k, v)


def _drop_in(name):
    """The reference's own settings files name ``ultra.*`` classes: they select the drop-in of the same name."""
    return 'ultra_amd.' + name[len('ultra.'):] if name.startswith('ultra.') else name


def _checkpoint_state(model_dir):
    index = os.path.join(model_dir, 'checkpoint')
    if not os.path.exists(index):
        return None
    with open(index) as fin:
        path = json.load(fin)['model_checkpoint_path']
    return path if os.path.exists(path + '.pt') else None


def _save_checkpoint(model, flags, exp_settings):
    os.makedirs(flags.model_dir, exist_ok=True)
    checkpoint_path = os.path.join(flags.model_dir, '%s.ckpt' % exp_settings['learning_algorithm'])
    full = model.saver.save(None, checkpoint_path, global_step=model.global_step)
    with open(os.path.join(flags.model_dir, 'checkpoint'), 'w') as fout:
        json.dump({'model_checkpoint_path': full}, fout)
    return full


def create_model(session, exp_settings, data_set, forward_only, flags):
    """main.py:56-75: the algorithm by dotted name, parameters from the last checkpoint if there is one."""
    model = find_class(_drop_in(exp_settings['learning_algorithm']))(data_set, exp_settings, forward_only)
    ckpt = _checkpoint_state(flags.model_dir)
    if ckpt:
        print('Reading model parameters from %s' % ckpt)
        model.saver.restore(session, ckpt)
    else:
        print('Created model with fresh parameters.')
    return model


def validate_model(model, data_set, data_input_feed, collect_scores=None):
    """main.py:153-165 (and :230-245): the whole set through get_next_batch, summaries merged by batch size."""
    it = 0
    summary_list, batch_size_list = [], []
    while it < len(data_set.initial_list):
        input_feed, info_map = data_input_feed.get_next_batch(it, data_set, check_validation=False)
        _, output_logits, summary = model.step(None, input_feed, True)
        summary_list.append(summary)
        batch_size_list.append(len(info_map['input_list']))
        if collect_scores is not None:
            for x in range(batch_size_list[-1]):
                collect_scores.append(output_logits[x])
        it += batch_size_list[-1]
    return merge_TFSummary(summary_list, batch_size_list)


def _load(flags, prefix, exp_settings):
    data = read_data(flags.data_dir, prefix, flags.max_list_cutoff)
    find_class(_drop_in(exp_settings['train_input_feed'])).preprocess_data(
        data, exp_settings['train_input_hparams'], exp_settings)
    return data


def train(exp_settings, flags=None):
    """main.py:78-202.  Returns the history: one dict per checkpoint interval (step, loss, valid/test metrics,
    the checkpoint written, if any)."""
    flags = flags or Flags()
    print('Reading data in %s' % flags.data_dir)
    train_set = _load(flags, flags.train_data_prefix, exp_settings)
    valid_set = _load(flags, flags.valid_data_prefix, exp_settings)
    print('Train Rank list size %d' % train_set.rank_list_size)
    print('Valid Rank list size %d' % valid_set.rank_list_size)
    exp_settings['max_candidate_num'] = max(train_set.rank_list_size, valid_set.rank_list_size)
    test_set = None
    if flags.test_while_train:
        test_set = _load(flags, flags.test_data_prefix, exp_settings)
        print('Test Rank list size %d' % test_set.rank_list_size)
        exp_settings['max_candidate_num'] = max(test_set.rank_list_size, exp_settings['max_candidate_num'])
        test_set.pad(exp_settings['max_candidate_num'])
    if 'selection_bias_cutoff' not in exp_settings:  # a limit on the number of items shown per training query
        exp_settings['selection_bias_cutoff'] = flags.selection_bias_cutoff if flags.selection_bias_cutoff > 0 \
            else exp_settings['max_candidate_num']
    exp_settings['selection_bias_cutoff'] = min(exp_settings['selection_bias_cutoff'], exp_settings['max_candidate_num'])
    print('Users can only see the top %d documents for each query in training.' % exp_settings['selection_bias_cutoff'])
    train_set.pad(exp_settings['max_candidate_num'])
    valid_set.pad(exp_settings['max_candidate_num'])

    print('Creating model...')
    model = create_model(None, exp_settings, train_set, False, flags)
    feed_of = lambda key: find_class(_drop_in(exp_settings[key + '_input_feed']))(  # noqa: E731
        model, flags.batch_size, exp_settings[key + '_input_hparams'], None)
    train_input_feed, valid_input_feed = feed_of('train'), feed_of('valid')
    test_input_feed = feed_of('test') if flags.test_while_train else None

    history = []
    step_time, loss = 0.0, 0.0
    current_step = 0
    best_perf = None
    pending = []
    while True:
        start_time = time.time()
        input_feed, _ = train_input_feed.get_batch(train_set, check_validation=True)
        step_loss, _, _ = model.step(None, input_feed, False)
        # The loss stays on the device until it is looked at, and looking at it waits for its step: the values are
        # read in groups of 32 (the result ring of a step holds 64), so the host keeps enqueueing ahead of the GPU.
        pending.append(step_loss)
        if len(pending) >= 32 or (current_step + 1) % flags.steps_per_checkpoint == 0:
            loss += sum(float(x) for x in pending) / flags.steps_per_checkpoint
            pending = []
        step_time += (time.time() - start_time) / flags.steps_per_checkpoint
        current_step += 1

        if current_step % flags.steps_per_checkpoint == 0:
            print('global step %d learning rate %.4f step-time %.2f loss %.4f' % (
                model.global_step.eval(), model.learning_rate.eval(), step_time, loss))
            entry = {'step': current_step, 'loss': loss, 'step_time': step_time, 'saved': None}
            valid_summary = validate_model(model, valid_set, valid_input_feed)
            print('  valid: %s' % ' '.join('%s:%.3f' % (x.tag, x.simple_value) for x in valid_summary.value))
            entry['valid'] = valid_summary.as_dict()
            if flags.test_while_train:
                test_summary = validate_model(model, test_set, test_input_feed)
                print('  test: %s' % ' '.join('%s:%.3f' % (x.tag, x.simple_value) for x in test_summary.value))
                entry['test'] = test_summary.as_dict()
            # the checkpoint is written when the objective metric on the validation set improves (:180-190)
            if 'objective_metric' in exp_settings:
                for x in valid_summary.value:
                    if x.tag == exp_settings['objective_metric']:
                        if current_step >= flags.start_saving_iteration:
                            if best_perf is None or best_perf < x.simple_value:
                                entry['saved'] = _save_checkpoint(model, flags, exp_settings)
                                best_perf = x.simple_value
                                print('Save model, valid %s:%.3f' % (x.tag, best_perf))
                                break
            # ... or at every interval when there is no objective metric (:191-194)
            if best_perf is None and current_step > flags.start_saving_iteration:
                entry['saved'] = _save_checkpoint(model, flags, exp_settings)
            history.append(entry)
            if loss == float('inf'):
                break
            step_time, loss = 0.0, 0.0
            sys.stdout.flush()
            if flags.max_train_iteration > 0 and current_step > flags.max_train_iteration:
                break
    return history


def test(exp_settings, flags=None):
    """main.py:206-262.  Returns (merged summary as a dict, scores per query)."""
    flags = flags or Flags()
    print('Reading data in %s' % flags.data_dir)
    test_set = _load(flags, flags.test_data_prefix, exp_settings)
    exp_settings['max_candidate_num'] = test_set.rank_list_size
    test_set.pad(exp_settings['max_candidate_num'])
    model = create_model(None, exp_settings, test_set, True, flags)
    test_input_feed = find_class(_drop_in(exp_settings['test_input_feed']))(
        model, flags.batch_size, exp_settings['test_input_hparams'], None)
    rerank_scores = []
    test_summary = validate_model(model, test_set, test_input_feed, collect_scores=rerank_scores)
    print('\n[Done]')
    print('  eval: %s' % ' '.join('%s:%.3f' % (x.tag, x.simple_value) for x in test_summary.value))
    if not os.path.exists(flags.output_dir):
        os.makedirs(flags.output_dir)
    output_ranklist(test_set, rerank_scores, flags.output_dir, flags.test_data_prefix)
    return test_summary.as_dict(), rerank_scores


def main(argv=None):
    defaults = Flags()
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    for name, value in sorted(vars(defaults).items()):
        if isinstance(value, bool):
            ap.add_argument('--' + name, type=lambda s: s.lower() in ('1', 'true', 'yes'), default=value)
        else:
            ap.add_argument('--' + name, type=type(value), default=value)
    flags = Flags(**vars(ap.parse_args(argv)))
    with open(flags.setting_file) as fin:
        exp_settings = json.load(fin)
    if flags.test_only:
        test(exp_settings, flags)
    else:
        train(exp_settings, flags)


if __name__ == '__main__':
    main()
