"""The reference's per-batch driver (/root/reference/train.py:17-80) on the MI355X path: same arguments, same
fetch list, same printed line, same 8-tuple -- a training script written against the reference keeps working
with ``from tspgnn.train import run_batch, summarize_epoch``."""
import numpy as np


def run_batch(sess, model, batch, batch_i, epoch_i, time_steps, train=False, verbose=True):
    EV, W, C, route_exists, n_vertices, n_edges = batch
    feed_dict = {
        model['EV']: EV, model['W']: W, model['C']: C, model['time_steps']: time_steps,
        model['route_exists']: route_exists, model['n_vertices']: n_vertices, model['n_edges']: n_edges,
    }
    outputs = [model['loss'], model['acc'], model['predictions'], model['TP'], model['FP'], model['TN'], model['FN']]
    if train:
        outputs = [model['train_step']] + outputs
    loss, acc, predictions, TP, FP, TN, FN = sess.run(outputs, feed_dict=feed_dict)[-7:]
    if verbose:
        print('{train_or_test} Epoch {epoch_i} Batch {batch_i}\t|\t(n,m,batch size)=({n},{m},{batch_size})\t|\t'
              '(Loss,Acc)=({loss:.4f},{acc:.4f})\t|\tAvg. (Sat,Prediction)=({avg_sat:.4f},{avg_pred:.4f})'.format(
                  train_or_test='Train' if train else 'Test', epoch_i=epoch_i, batch_i=batch_i, loss=loss, acc=acc,
                  n=np.sum(n_vertices), m=np.sum(n_edges), batch_size=n_vertices.shape[0],
                  avg_sat=np.mean(route_exists), avg_pred=np.mean(np.round(predictions))), flush=True)
    return loss, acc, np.mean(route_exists), np.mean(predictions), TP, FP, TN, FN


def summarize_epoch(epoch_i, loss, acc, sat, pred, train=False):
    print('{train_or_test} Epoch {epoch_i} Average\t|\t(Loss,Acc)=({loss:.4f},{acc:.4f})\t|\t'
          'Avg. (Sat,Pred)=({avg_sat:.4f},{avg_pred:.4f})'.format(
              train_or_test='Train' if train else 'Test', epoch_i=epoch_i, loss=np.mean(loss), acc=np.mean(acc),
              avg_sat=np.mean(sat), avg_pred=np.mean(pred)), flush=True)
